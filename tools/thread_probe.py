"""Which thread of a bench.py process burns host CPU, and where: starts `python bench.py …` as a child process, then
samples /proc/<pid>/task/*/{stat,syscall} of the child for a few seconds (no ptrace, no debugger) and prints per thread the
CPU time used, the histogram of system calls it was found in ("running" = user space) and the shared objects its
program counter fell into when it was inside a system call.

  python tools/thread_probe.py [--delay 30] [--window 4] -- --steps 400 --warmup 20 --no-cpu-baseline --no-other-configs
"""
import argparse
import bisect
import collections
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def maps(pid):
    out = []
    with open(f"/proc/{pid}/maps") as f:
        for ln in f:
            p = ln.split()
            if len(p) >= 6 and "x" in p[1]:
                lo, hi = (int(v, 16) for v in p[0].split("-"))
                out.append((lo, hi, int(p[2], 16), p[5]))
    out.sort()
    return out


def locate(mp, pc):
    i = bisect.bisect_right([m[0] for m in mp], pc) - 1
    if i >= 0 and mp[i][0] <= pc < mp[i][1]:
        return f"{os.path.basename(mp[i][3])}+0x{pc - mp[i][0] + mp[i][2]:x}"
    return hex(pc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--delay", type=float, default=30.0)
    ap.add_argument("--window", type=float, default=4.0)
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    rest = [r for r in a.rest if r != "--"]
    child = subprocess.Popen([sys.executable, "bench.py"] + rest, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True)
    time.sleep(a.delay)
    pid = child.pid
    tick = os.sysconf("SC_CLK_TCK")

    def cpu():
        out = {}
        for tid in os.listdir(f"/proc/{pid}/task"):
            try:
                with open(f"/proc/{pid}/task/{tid}/stat") as f:
                    s = f.read()
                name = s[s.index("(") + 1:s.rindex(")")]
                fld = s[s.rindex(")") + 2:].split()
                out[tid] = (name, (int(fld[11]) + int(fld[12])) / tick)
            except OSError:
                pass
        return out
    c0, t0 = cpu(), time.time()
    sysc = collections.defaultdict(collections.Counter)
    pcs = collections.defaultdict(collections.Counter)
    mp = maps(pid)
    while time.time() - t0 < a.window:
        for tid in c0:
            try:
                with open(f"/proc/{pid}/task/{tid}/syscall") as f:
                    p = f.read().split()
            except OSError:
                continue
            if not p:
                continue
            sysc[tid][p[0]] += 1
            if p[0] not in ("running", "-1") and len(p) >= 9:
                pcs[tid][locate(mp, int(p[8], 16))] += 1
        time.sleep(0.002)
    c1, dt = cpu(), time.time() - t0
    for tid in sorted(c1, key=lambda t: -(c1[t][1] - c0.get(t, (0, 0))[1])):
        used = c1[tid][1] - c0.get(tid, (0, 0))[1]
        if used < 0.02 * dt:
            continue
        print(f"tid {tid} {c1[tid][0]}: {100 * used / dt:.0f} % of a core; syscalls {dict(sysc[tid].most_common(6))}; "
              f"pc {dict(pcs[tid].most_common(6))}")
    out, err = child.communicate(timeout=900)
    print(err[-1500:])
    print(out[-300:])


if __name__ == "__main__":
    main()
