"""Sampling profiler of the Python threads of a script (no ptrace, no extension): a daemon thread reads
``sys._current_frames()`` every millisecond and counts, per thread, the source LINE that is executing (for a thread inside
a C call — a ctypes entry point, a torch operator — that is the calling line) and every function on its stack.

  python tools/py_sampler.py [--skip 12] [--top 45] bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-other-configs
"""
import collections
import os
import runpy
import sys
import threading
import time


def main():
    argv = sys.argv[1:]
    skip, top = 12.0, 45
    while argv and argv[0] in ("--skip", "--top"):
        if argv[0] == "--skip":
            skip = float(argv[1])
        else:
            top = int(argv[1])
        argv = argv[2:]
    script = argv[0]
    sys.argv = argv
    lines = collections.defaultdict(collections.Counter)
    funcs = collections.defaultdict(collections.Counter)
    total = collections.Counter()
    stop = []
    me = []

    def sampler():
        me.append(threading.get_ident())
        t0 = time.time()
        while not stop:
            time.sleep(0.001)
            if time.time() - t0 < skip:
                continue
            for tid, fr in sys._current_frames().items():
                if tid == me[0]:
                    continue
                total[tid] += 1
                f = fr
                lines[tid][(os.path.basename(f.f_code.co_filename), f.f_lineno, f.f_code.co_name)] += 1
                seen = set()
                while f is not None:
                    key = (os.path.basename(f.f_code.co_filename), f.f_code.co_name)
                    if key not in seen:
                        funcs[tid][key] += 1
                        seen.add(key)
                    f = f.f_back
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        stop.append(1)
        names = {t.ident: t.name for t in threading.enumerate()}
        for tid in sorted(total, key=lambda t: -total[t]):
            n = total[tid]
            if n < 50:
                continue
            print(f"\n=== thread {names.get(tid, tid)}: {n} samples", file=sys.stderr)
            print("--- lines (self)", file=sys.stderr)
            for (fn, ln, name), c in lines[tid].most_common(top):
                print(f"{100 * c / n:6.2f} %  {fn}:{ln} {name}", file=sys.stderr)
            print("--- functions (on stack)", file=sys.stderr)
            for (fn, name), c in funcs[tid].most_common(top):
                print(f"{100 * c / n:6.2f} %  {fn} {name}", file=sys.stderr)


if __name__ == "__main__":
    main()
