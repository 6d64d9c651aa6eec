import ctypes as C, time, sys
sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg=load_package(); lib=pkg.lib()
lib.bmh_ctx_create.argtypes=[C.POINTER(C.c_void_p), C.c_int]
lib.bmh_ctx_reserve_staging.argtypes=[C.c_void_p, C.c_size_t, C.c_size_t]
lib.bmh_ctx_reserve_device.argtypes=[C.c_void_p, C.c_size_t, C.c_int64, C.c_size_t]
for mb in (0, 16, 64):
    for k in range(4):
        h=C.c_void_p()
        t0=time.time(); rc=lib.bmh_ctx_create(C.byref(h),0); t1=time.time()
        if mb: lib.bmh_ctx_reserve_staging(h, mb<<20, mb<<20)
        t2=time.time()
        if mb: lib.bmh_ctx_reserve_device(h, mb<<20, (mb<<10)*2, (mb<<20)//8)
        t3=time.time()
        print(f"reserve {mb} MB: ctx {k}: create {1e3*(t1-t0):.1f} ms, staging {1e3*(t2-t1):.1f} ms, device {1e3*(t3-t2):.1f} ms", flush=True)
