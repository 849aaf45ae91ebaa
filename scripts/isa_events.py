#!/usr/bin/env python
"""Compact event trace of a kernel's ISA (hipcc -S --cuda-device-only output): runs of MFMAs, loads, waits, exec-mask blocks and barriers
in program order -- the view in which a prefetch that the compiler serialised (s_waitcnt vmcnt(N) between the loads of one chunk, loads
under s_and_saveexec) shows up at a glance.   usage: python scripts/isa_events.py file.s [kernel-name-substring]"""
import re
import sys

def main():
    path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    name, ev = None, []
    def flush():
        if name and pat in name and any(e[0] == "mfma" for e in ev):
            print("==", name)
            out, run = [], 0
            for kind, txt in ev:
                if kind == "mfma":
                    run += 1
                    continue
                if run:
                    out.append(f"M{run}")
                    run = 0
                out.append(txt)
            if run:
                out.append(f"M{run}")
            # collapse repeats
            line, prev, cnt = [], None, 0
            for t in out + [None]:
                if t == prev:
                    cnt += 1
                    continue
                if prev is not None:
                    line.append(prev if cnt == 1 else f"{prev}x{cnt}")
                prev, cnt = t, 1
            print(" ".join(line))
    for ln in open(path):
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            flush()
            name, ev = m.group(1), []
            continue
        if name is None:
            continue
        if s.startswith("v_mfma") or s.startswith("v_smfma"):
            ev.append(("mfma", ""))
        elif s.startswith("global_load") or s.startswith("buffer_load"):
            ev.append(("x", "L" + ("lds" if " lds" in s else "")))
        elif s.startswith("global_store"):
            ev.append(("x", "S"))
        elif s.startswith("s_waitcnt") and "vmcnt" in s:
            ev.append(("x", "w" + re.search(r"vmcnt\((\d+)\)", s).group(1)))
        elif s.startswith("s_and_saveexec"):
            ev.append(("x", "{"))
        elif s.startswith("s_barrier"):
            ev.append(("x", "|B|"))
        elif s.startswith("ds_write") or s.startswith("ds_read"):
            ev.append(("x", "dw" if s.startswith("ds_write") else "dr"))
        elif re.match(r"^\.LBB\d+_\d+:.*Loop Header", s):
            ev.append(("x", "\n  [loop " + s.split(":")[0] + "]"))
        elif s.startswith("s_cbranch"):
            ev.append(("x", "br"))
    flush()
main()
