"""in-process (torch loaded) step-by-step probe of an external event record inside a stream capture, through ctypes on the HIP
runtime torch brought in"""
import ctypes, torch
torch.zeros(1, device='cuda')
import os
path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
hip = ctypes.CDLL(path)
hip.hipGetErrorString.restype = ctypes.c_char_p
def ck(name, rc):
    print('%-40s -> %d %s' % (name, rc, hip.hipGetErrorString(rc).decode()))
    return rc
for mode_name, mode in (('global', 0), ('threadlocal', 1), ('relaxed', 2)):
    for sflag in (0, 1):
        s = ctypes.c_void_p(); ev = ctypes.c_void_p(); g = ctypes.c_void_p()
        ck('streamCreate flags=%d' % sflag, hip.hipStreamCreateWithFlags(ctypes.byref(s), sflag))
        ck('eventCreate', hip.hipEventCreateWithFlags(ctypes.byref(ev), 2))
        ck('beginCapture ' + mode_name, hip.hipStreamBeginCapture(s, mode))
        ck('recordWithFlags external', hip.hipEventRecordWithFlags(ev, s, 1))
        ck('endCapture', hip.hipStreamEndCapture(s, ctypes.byref(g)))
        hip.hipGetLastError()
# torch capture
st = torch.cuda.Stream()
ev = ctypes.c_void_p(); hip.hipEventCreateWithFlags(ctypes.byref(ev), 2)
gr = torch.cuda.CUDAGraph()
a = torch.zeros(1024, device='cuda')
with torch.cuda.graph(gr):
    a.add_(1)
    rc = hip.hipEventRecordWithFlags(ev, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 1)
    a.add_(1)
print('inside torch.cuda.graph: record ->', rc, hip.hipGetErrorString(rc).decode())
hip.hipGetLastError()
