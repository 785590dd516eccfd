"""pytest against an experimental build of the library: python tools/pytest_with_lib.py <lib.so> [pytest args]."""
import os, sys
sys.path.insert(0, ".")
import vision_amd._abi as abi
abi.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
