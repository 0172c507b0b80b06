import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the emulator runs one workgroup per OS thread at a time; kernels whose workgroups wait for each other (the grouped
# sampler tail) need at least as many emulator threads as workgroups that wait together (<= 8), whatever the core count of the box
os.environ.setdefault("HIPEMU_THREADS", str(max(8, os.cpu_count() or 1)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emu: runs the HIP kernel sources under the CPU SIMT emulator (dev tool)")
