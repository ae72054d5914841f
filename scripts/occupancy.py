import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch
from gangealing_amd import _lib
lib = _lib.load()
torch.zeros(1, device='cuda')
buf = ctypes.create_string_buffer(2048)
lib.gg_debug_conv_occupancy.argtypes = [ctypes.c_char_p, ctypes.c_int]
print(lib.gg_debug_conv_occupancy(buf, 2048))
print(buf.value.decode().replace(';', '\n'))
