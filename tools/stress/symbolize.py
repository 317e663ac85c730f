"""Resolve the module(+offset) frames of a crash_bt dump to the nearest exported / local symbols:

    python tools/stress/symbolize.py gpurun_out/stress_r3_hogs.txt

(nm on the modules of THIS image -- the GPU boxes run the same image; modules given by a path relative
to the repository are looked up from its root.)"""
import bisect
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_tables = {}


def table(path):
    if path not in _tables:
        syms = []
        for flags in (["-D", "--defined-only"], ["--defined-only"]):
            try:
                out = subprocess.run(["nm", "-C"] + flags + [path], capture_output=True, text=True).stdout
            except Exception:
                out = ""
            for ln in out.splitlines():
                parts = ln.split(None, 2)
                if len(parts) == 3 and parts[1] in "TtWwiV":
                    try:
                        syms.append((int(parts[0], 16), parts[2]))
                    except ValueError:
                        pass
        syms = sorted(set(syms))
        _tables[path] = ([a for a, _ in syms], [n for _, n in syms])
    return _tables[path]


def main():
    for ln in open(sys.argv[1]):
        m = re.match(r"^(\S+?)\((\S*?)\+0x([0-9a-f]+)\)\[", ln.strip())
        if not m:
            if "crash_bt" in ln:
                print(ln.rstrip())
            continue
        mod, sym, off = m.group(1), m.group(2), int(m.group(3), 16)
        path = mod if os.path.isabs(mod) else os.path.join(ROOT, mod)
        if sym or not os.path.exists(path):
            print(f"  {os.path.basename(mod)}: {sym}+0x{off:x}")
            continue
        addrs, names = table(path)
        i = bisect.bisect_right(addrs, off) - 1
        # (a stripped module: the nearest exported symbol far below the address says nothing)
        where = f"{names[i]}+0x{off - addrs[i]:x}" if i >= 0 and off - addrs[i] < 0x3000 else f"+0x{off:x} (internal, stripped)"
        print(f"  {os.path.basename(mod)}: {where}")


if __name__ == "__main__":
    main()
