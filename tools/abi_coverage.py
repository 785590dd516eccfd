"""Which entry points of include/detgeom.h does `pytest -m gpu` actually call?  Runs the GPU tests in this process with the
library behind a proxy that notes every symbol fetched from it, then lists the declared symbols that no test reached.
usage (GPU box): python tools/abi_coverage.py [pytest args]"""
import sys
sys.path.insert(0, ".")
import pytest
import vision_amd._abi as abi

real = abi.load()
used = set()

class Proxy:
    def __getattr__(self, name):
        used.add(name)
        return getattr(real, name)

abi._lib = Proxy()
rc = pytest.main(["tests", "-m", "gpu", "-x", "-q", "--timeout", "150", "--timeout-method", "thread", "-p", "no:cacheprovider"] + sys.argv[1:])
missing = sorted(set(abi.SIGNATURES) - used)
print(f"pytest rc {rc}; {len(used & set(abi.SIGNATURES))} of {len(abi.SIGNATURES)} declared entry points called; not called: {missing}")
