"""The same launch loop from a Python process with torch loaded (ctypes -> a tiny shared library)."""
import ctypes, sys, os
import torch
torch.zeros(1, device="cuda")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "liblaunch_rate.so"))
lib.run()
