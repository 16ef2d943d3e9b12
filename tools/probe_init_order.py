"""Probe: does torch initialise after librt_amd.so's rt_init (and the other way round)?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
lib_path = sys.argv[2] if len(sys.argv) > 2 else None
def ours():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("PROBE_VIA_CAPI"):
        from raytracer_amd import capi
        capi._share_the_process_hip_runtime()
    lib = C.CDLL(lib_path or os.path.join(here, "raytracer_amd", "csrc", "librt_amd.so"))
    lib.rt_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.rt_last_error.argtypes = [C.c_void_p]; lib.rt_last_error.restype = C.c_char_p
    ctx = C.c_void_p()
    rc = lib.rt_init(0, C.byref(ctx))
    print("rt_init rc", rc, lib.rt_last_error(None).decode(), flush=True)
    return lib, ctx
def theirs():
    import torch
    print("torch device_count", torch.cuda.device_count(), flush=True)
    try:
        torch.cuda.init(); x = torch.zeros(4, device="cuda"); print("torch ok", x.sum().item(), flush=True)
    except Exception as e:
        print("torch FAILED:", e, flush=True)
if order == "ours_first":
    ours(); theirs()
else:
    theirs(); ours()
