#!/usr/bin/env python
"""ISA lint for kernels that hide global loads from the compiler (inline-asm ``global_load_dwordx4`` into registers that are consumed
tens of instructions later behind a hand-counted ``s_waitcnt vmcnt(N)``): between the load and the wait that retires it, NO instruction
may read or write the destination registers -- a compiler-inserted copy there would move stale bytes.  Walks the straight-line code
of every kernel whose name matches ``--kernel`` in a gfx950 assembly listing (``hipcc --save-temps`` / ``-S``) and replays the
vector-memory queue: loads and LDS-DMAs retire in order, ``vmcnt(N)`` leaves the N youngest outstanding.

  python tools/isa_lint_inflight.py conv_gemm-hip-amdgcn-amd-amdhsa-gfx950.s --kernel conv_nt3_kernel

Exit code 1 and a report on the first violation per kernel.  Forward branches are replayed as straight-line text; a backward branch
(a loop) with hidden loads in flight is reported."""
import argparse
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint(lines, name):
    queue = []      # outstanding VMEM ops, oldest first: (line_no, set of destination VGPRs or empty)
    lq = []         # outstanding LDS ops (lgkmcnt): the same replay for hand-counted ds_read_* (they return in order)
    problems = []
    labels, headers, last_label = {}, set(), None
    for no, raw in lines:
        m = re.match(r"^(\.?\w+):", raw.strip())
        if m:
            labels[m.group(1)] = no
            last_label = (m.group(1), no)
        if "Loop Header" in raw and last_label is not None and no - last_label[1] <= 1:
            headers.add(last_label[0])   # hipcc annotates loop header blocks: a branch to one of THEM is a loop back edge
    in_asm = False  # inside a ;;#ASMSTART ... ;;#ASMEND bracket: only THOSE loads are hidden from the compiler's own wait insertion
    for no, raw in lines:
        if "#ASMSTART" in raw:
            in_asm = True
        elif "#ASMEND" in raw:
            in_asm = False
        ins = raw.split(";")[0].strip()
        if not ins or ins.endswith(":") or ins.startswith("."):
            continue
        op = ins.split()[0]
        if op.startswith("global_load_lds") or op.startswith("buffer_load") and " lds" in ins:
            queue.append((no, frozenset()))
            continue
        if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
            dst = regs_of(ins.split(",")[0]) if in_asm else set()   # compiler-issued loads: the compiler waits for them itself
            queue.append((no, frozenset(dst)))
            continue
        if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store") or op.startswith("global_atomic"):
            queue.append((no, frozenset()))   # stores count on vmcnt too (gfx9): conservative, they carry no destination
        if op.startswith("ds_"):
            dst = regs_of(ins.split(",")[0]) if (in_asm and op.startswith("ds_read")) else set()
            lq.append((no, frozenset(dst)))
            if not dst:
                continue
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            if any(d for _, d in lq):
                problems.append("%s: line %d `%s`: scalar load while hand-counted LDS reads are in flight (lgkmcnt is shared)" % (name, no, ins))
                break
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                keep = int(m.group(1))
                if keep < len(queue):
                    queue = queue[len(queue) - keep:] if keep else []
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                keep = int(m.group(1))
                if keep < len(lq):
                    lq = lq[len(lq) - keep:] if keep else []
            continue
        inflight = set()
        for _, d in queue:
            inflight |= d
        for no_, d in lq:
            if no_ != no:
                inflight |= d
        if inflight:
            touched = regs_of(ins) & inflight
            if touched:
                src = [q for q in queue + lq if q[1] & touched]
                problems.append("%s: line %d `%s` touches v%s while the load of line %d is in flight"
                                % (name, no, ins, sorted(touched), src[0][0]))
                break
            if op.startswith("s_cbranch") or op == "s_branch":
                # a FORWARD branch is replayed as straight-line text (both arms are checked for touches, which is conservative; no
                # register of a load in flight can need a phi copy unless an arm writes it, and that is a touch); a branch BACK to a
                # loop header with loads in flight is the loop-carried case that made hipcc copy registers before their data had landed
                # (backward branches to other blocks are the compiler's layout of if / else joins inside one iteration)
                target = ins.split()[-1]
                if labels.get(target, 1 << 60) <= no and (target in headers or not headers):
                    problems.append("%s: line %d `%s`: backward branch while asm loads are in flight" % (name, no, ins))
                    break
    return problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing")
    ap.add_argument("--kernel", default="conv_nt3_kernel")
    a = ap.parse_args()
    text = open(a.listing).read().splitlines()
    kernels, cur, name = {}, None, None
    for i, line in enumerate(text, 1):
        m = re.match(r"^(_Z\w+):", line)
        if m and a.kernel in m.group(1):
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is not None:
            cur.append((i, line))
            if "s_endpgm" in line:
                cur = None
    if not kernels:
        print("no kernel matching %r in %s" % (a.kernel, a.listing))
        return 2
    bad = []
    for name, lines in kernels.items():
        bad += lint(lines, name)
    for b in bad:
        print(b)
    print("%d kernel(s) checked, %d problem(s)" % (len(kernels), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
