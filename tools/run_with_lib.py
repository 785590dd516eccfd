"""Run a tool script against an experimental build of the library: python tools/run_with_lib.py <lib.so> <script.py> [args]."""
import os, runpy, sys
sys.path.insert(0, ".")
import vision_amd._abi as abi
abi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
