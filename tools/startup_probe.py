"""Developer probe of the GPU box's fixed costs: HIP bring-up through libfqtk_match.so, page-locking memory, and creating
the 771 output files of a 384-sample run on RAM-backed scratch -- alone, from several threads, and side by side."""
import ctypes as C, os, shutil, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "seq"
d = "/dev/shm/fqtk_probe_files"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)

def create(lo, hi):
    fds = [os.open(f"{d}/f{i}.fq.gz", os.O_WRONLY | os.O_CREAT | os.O_TRUNC) for i in range(lo, hi)]
    for f in fds:
        os.close(f)

def hip():
    t1 = time.perf_counter()
    lib = C.CDLL(os.path.join(ROOT, "fqtk_amd", "lib", "libfqtk_match.so"))
    p = C.c_void_p()
    lib.fqtk_pinned_alloc(C.c_size_t(1 << 20), C.byref(p))
    print(f"  first pinned alloc done after {time.perf_counter() - t1:.3f} s")

t0 = time.perf_counter()
if mode == "seq":
    create(0, 771)
    print(f"771 files, one thread: {time.perf_counter() - t0:.3f} s")
elif mode == "par":
    th = [threading.Thread(target=create, args=(k * 193, min(771, (k + 1) * 193))) for k in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    print(f"771 files, four threads: {time.perf_counter() - t0:.3f} s")
elif mode == "both":
    h = threading.Thread(target=hip); h.start()
    create(0, 771)
    print(f"771 files next to the HIP bring-up: {time.perf_counter() - t0:.3f} s")
    h.join()
elif mode == "tmp":
    d = "/tmp/fqtk_probe_files"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
    create(0, 771)
    print(f"771 files in /tmp: {time.perf_counter() - t0:.3f} s")
shutil.rmtree(d, ignore_errors=True)
