"""tools/prof_nms.py against an experimental library: PN_LIB=<lib.so> (rocprofv3 wants the program itself after --)."""
import os, sys, runpy
sys.path.insert(0, ".")
import vision_amd._abi as abi
if os.environ.get("PN_LIB"): abi.LIB_PATH = os.path.abspath(os.environ["PN_LIB"])
runpy.run_path("tools/prof_nms.py", run_name="__main__")
