"""hehub_amd -- MI355X-native RNS polynomial-ring engine behind hehub's hot-path API.

Contents (only what the path needs):
  csrc/     HIP kernels (gfx950) + the C-ABI implementation (include/hehub_amd.h)
  host/     C++ host layer mirroring hehub's RnsPolynomial / ckks:: / bgv:: interface
  capi.py   ctypes declarations of the C ABI
  engine.py thin Python front-end used by tests and bench.py (torch = device buffers only)
  dist.py   batch sharding across ranks (rendezvous, fences; no collective on the data path)
  sharded.py  limb-sharded latency mode: one operation cut by output modulus, exchanges over torch.distributed
  wire.py   ctypes wrappers of the wire / on-disk format
  build.py  in-tree hipcc build of lib/libhehub_amd.so
"""
__version__ = "0.1.0"
