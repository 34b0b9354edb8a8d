"""Does hipIpc work between two processes that share GPU 0 on this box?  A allocates + fills a buffer and exports it; B opens the
handle, checks the bytes, writes back; A verifies.  (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0.)"""
import ctypes as C, os, subprocess, sys, time
hip = C.CDLL("libamdhip64.so")
def chk(rc, what):
    if rc != 0: raise SystemExit("%s failed: %d" % (what, rc))
N = 1 << 20
if len(sys.argv) == 1:
    if os.environ.get("PROBE_PTRACER") == "1":
        libc = C.CDLL("libc.so.6"); print("prctl rc", libc.prctl(0x59616d61, C.c_ulong(-1 & 0xFFFFFFFFFFFFFFFF), 0, 0, 0))   # PR_SET_PTRACER, PR_SET_PTRACER_ANY
    p = C.c_void_p(); chk(hip.hipMalloc(C.byref(p), N), "hipMalloc")
    host = (C.c_ubyte * N)(*([7] * N)); chk(hip.hipMemcpy(p, host, N, 1), "h2d")
    h = (C.c_ubyte * 64)(); chk(hip.hipIpcGetMemHandle(C.byref(h), p), "hipIpcGetMemHandle")
    open("/tmp/ipc_handle.bin", "wb").write(bytes(h))
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, timeout=120)
    print("child:", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
    chk(hip.hipDeviceSynchronize(), "sync")
    chk(hip.hipMemcpy(host, p, N, 2), "d2h")
    print("parent sees", host[0], host[N - 1], "(expect 9 9)")
else:
    raw = open("/tmp/ipc_handle.bin", "rb").read()
    h = (C.c_ubyte * 64).from_buffer_copy(raw)
    q = C.c_void_p()
    rc = hip.hipIpcOpenMemHandle(C.byref(q), h, 1)
    print("open rc", rc, end=" ")
    if rc == 0:
        host = (C.c_ubyte * N)(); chk(hip.hipMemcpy(host, q, N, 2), "d2h")
        print("child sees", host[0], host[N - 1], end=" ")
        host2 = (C.c_ubyte * N)(*([9] * N)); chk(hip.hipMemcpy(q, host2, N, 1), "h2d"); chk(hip.hipDeviceSynchronize(), "sync")
        chk(hip.hipIpcCloseMemHandle(q), "close")
