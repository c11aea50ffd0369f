import multiprocessing as mp, time, os
def spin(t):
    n=0; e=time.perf_counter()+t
    while time.perf_counter()<e:
        for _ in range(10000): n+=1
    return n
if __name__=="__main__":
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
    print("cpus_allowed:", [l for l in open("/proc/self/status") if l.startswith("Cpus_allowed_list")][0].strip(), "affinity", len(os.sched_getaffinity(0)))
    base=spin(1.0)
    for p in (1,8,16,32,64,128):
        with mp.Pool(p) as pool:
            t=time.perf_counter(); r=pool.map(spin,[1.0]*p); dt=time.perf_counter()-t
        print(p, "procs: total/1-proc =", round(sum(r)/base,1), "wall", round(dt,2))
