import sys, time, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goleft_amd import _hostlib as hl
lib=hl.load()
for th in (64,):
    b=C.c_void_p()
    t0=time.perf_counter()
    assert lib.gdh_bam_open(b"/tmp/gd_s3.bam", th, C.byref(b))==0
    tid=C.c_int32(); n=C.c_size_t(); m=C.c_size_t(); p=[C.c_void_p() for _ in range(5)]
    tot=0
    while True:
        rc=lib.gdh_bam_next(b, 1<<21, C.byref(tid), C.byref(n), C.byref(m), *[C.byref(x) for x in p])
        if rc<=0: break
        tot+=n.value
    dt=time.perf_counter()-t0
    lib.gdh_bam_close(b)
    print("threads",th,"records",tot,"%.3f s"%dt)
