#!/usr/bin/env python
"""Do memcpy / memset nodes of a captured hipGraph replay?  torch's contiguous copy_ (hipMemcpyAsync D2D) and the library-style hipMemsetAsync, captured and
replayed three times on changing sources."""
import ctypes
import torch
d = torch.device('cuda', 0)
src = torch.randn(1 << 20, device=d)
dst = torch.zeros(1 << 20, device=d)
z = torch.ones(1 << 20, device=d)
hip = ctypes.CDLL('libamdhip64.so')
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    g.capture_begin()
    dst.copy_(src)                                                       # memcpy node
    hip.hipMemsetAsync(ctypes.c_void_p(z.data_ptr()), 0, ctypes.c_size_t(z.numel() * 4), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))      # memset node
    g.capture_end()
torch.cuda.current_stream().wait_stream(s)
for rep in range(3):
    src.normal_()
    z.fill_(1.0)
    dst.fill_(-1.0)
    g.replay()
    torch.cuda.synchronize()
    print('replay %d: copy_ node copied: %s   memset node zeroed: %s' % (rep, bool(torch.equal(dst, src)), bool((z == 0).all())))
