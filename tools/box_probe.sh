df -h /dev/shm /tmp | cat; free -g; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null; ulimit -l; ulimit -n; rocm-smi --showmeminfo vram | head -8
