"""Static lint of the generated gfx950 ISA for the hazards the compiler cannot place wait states for.

libskm's hot kernels issue their MFMAs from `asm volatile` blocks (AGPR-resident accumulators, hand-scheduled operand
streams).  LLVM's hazard recogniser does not look inside inline asm, so the wait states between such an MFMA and the
instructions around it are placed by hand (`s_nop`s in csrc/skm_gmm_kernels.h) -- and a compiler point release may move
a copy, a spill or a clear next to one of them.  This tool disassembles the device code of a library or object file
and checks, per kernel, linearly over the instruction stream:

  (i)   MFMA result -> next use.  A non-MFMA instruction that reads or writes a register of an MFMA's destination, or a
        later MFMA that reads it as SrcA / SrcB (or as a SrcC that is not the very same tuple), fewer than W wait states
        behind it.  W as in LLVM's GCNHazardRecognizer for gfx940 / gfx950: passes + 2 for the float32 MFMAs
        (32x32x2: 18, 16x16x4: 10), 11 for the float64 16x16x4 (18 when a store reads it), passes + 4 for the
        bf16 32x32x16 (12).
  (ii)  VALU write -> MFMA read.  A vector-ALU instruction (v_accvgpr_write / v_mov included) that writes a register an
        MFMA reads fewer than 2 wait states later.
  (iii) LDS-direct loads.  `buffer_load ... lds` whose data a `ds_read` of the NEXT loop iteration (or, outside a loop,
        any later `ds_read`) can reach before an `s_waitcnt vmcnt(n)` that covers the load: hipcc emits no such wait
        for this load kind.  (A `ds_read` in the SAME iteration belongs to the other buffer of the ring by design.)
        Kernels listed in RING_DEPTH keep their loads in flight longer -- a ring of three LDS stages filled two
        iterations ahead: there the reads of the iteration after next are the ones that must be covered.

A wait state is one issued instruction; `s_nop n` counts n + 1.  One refinement over LLVM's counting (which is one per
instruction whatever it is): the matrix pipe takes one MFMA at a time, so a second MFMA of the same kind issues no
sooner than `passes` wait states behind the first -- behind it the first counts as that old.  The kernels rely on
exactly this ("only the last MFMA of a group can still be in flight"); `--strict` switches it off.  Windows follow both arms of every branch (the control
flow graph is rebuilt from the branch offsets; "next iteration" means behind a back edge of a natural loop that holds
the load).  Registers are compared by number: the lint knows nothing of the values, so it reports a pair the program
can never execute together as readily as a real one.

  python tools/isa_lint.py [file ...]      (default: speaker-recognition_amd/libskm.so)     exit status 1 = findings
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("SKM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# destination -> other instruction: wait states needed (LLVM GCNHazardRecognizer, gfx940 / gfx950 tables)
MFMA_WAIT = [
    (re.compile(r"^v_mfma_f32_32x32x2_?f32"), 18, 18),
    (re.compile(r"^v_mfma_f32_16x16x4_?f32"), 10, 10),
    (re.compile(r"^v_mfma_f32_4x4x1"), 4, 4),
    (re.compile(r"^v_mfma_f64_16x16x4_?f64"), 11, 18),
    (re.compile(r"^v_mfma_f64_4x4x4"), 6, 9),
    (re.compile(r"^v_mfma_f32_32x32x16_"), 12, 12),
    (re.compile(r"^v_mfma_f32_16x16x32_"), 8, 8),
    (re.compile(r"^v_mfma_"), 19, 19),          # anything else: the longest XDL case
]
# passes (4 clocks each) the matrix pipe is busy with one instruction
MFMA_PASSES = [
    (re.compile(r"^v_mfma_f32_32x32x2_?f32"), 16),
    (re.compile(r"^v_mfma_f32_16x16x4_?f32"), 8),
    (re.compile(r"^v_mfma_f64_16x16x4_?f64"), 8),
    (re.compile(r"^v_mfma_f32_32x32x16_"), 8),
]
STRICT = False
# LDS-direct loads: iterations between a load and the first ds_read of its data (default 1: a ring of two buffers)
RING_DEPTH = [
    (re.compile(r"k_gemm_b3"), 2),      # csrc/skm_gemm_b3.hip: three stages, loads issued two stages ahead, vmcnt(6)
]
REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])(?![\w\[])")
LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
LABEL = re.compile(r"^[0-9a-f]+ <(.+)>:")


def regs(text):
    out = []
    for m in REG.finditer(text):
        f = m.group(1)
        if m.group(2) is not None:
            lo = hi = int(m.group(2))
        else:
            lo, hi = int(m.group(3)), int(m.group(4))
        out.append((f, lo, hi))
    return out


def overlap(a, b):
    return a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


class Ins(object):
    __slots__ = ("op", "args", "addr", "ops", "ws", "is_mfma", "is_valu", "vmem", "lds_direct", "is_store")

    def __init__(self, op, args, addr):
        self.op, self.args, self.addr = op, args, addr
        self.ops = [regs(a) for a in split_args(args)]
        self.ws = 1
        if op == "s_nop":
            try:
                self.ws = int(args.strip(), 0) + 1
            except ValueError:
                pass
        self.is_mfma = op.startswith("v_mfma") or op.startswith("v_smfma")
        self.is_valu = op.startswith("v_") and not self.is_mfma
        self.vmem = op.startswith(("buffer_", "global_", "flat_", "scratch_"))
        self.lds_direct = self.vmem and "load" in op and re.search(r"\blds\b", args) is not None
        self.is_store = ("store" in op) or op.startswith("ds_write") or op.startswith("ds_store")


def split_args(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def disassemble(path):
    """{kernel name: [Ins]} of every gfx950 code object bundled in `path`."""
    tmp = tempfile.mkdtemp(prefix="isa_lint_")
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not objs:                                    # a bare code object
            objs = [os.path.basename(local)]
        kernels = {}
        for o in objs:
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, o)], check=True,
                                 capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = LABEL.match(line)
                if m:
                    cur = kernels.setdefault(m.group(1), [])
                    continue
                m = LINE.match(line)
                if m and cur is not None:
                    cur.append(Ins(m.group(1), m.group(2), int(m.group(3), 16)))
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def mfma_passes(op):
    for rx, p in MFMA_PASSES:
        if rx.match(op):
            return p
    return 0


def mfma_wait(op):
    for rx, w, wmem in MFMA_WAIT:
        if rx.match(op):
            return w, wmem
    return 19, 19


def branch_target(b):
    try:
        simm = int(b.args.strip().split()[0], 0)
    except (ValueError, IndexError):
        return None
    if simm >= 0x8000:
        simm -= 0x10000
    return b.addr + 4 + 4 * simm


class Flow(object):
    """Instruction-level successors, basic blocks, dominators and the natural loops of one kernel."""

    def __init__(self, ins):
        n = len(ins)
        self.ins = ins
        index = {a.addr: k for k, a in enumerate(ins)}
        self.succ = [()] * n
        leaders = {0}
        for i, a in enumerate(ins):
            if a.op == "s_branch" or a.op.startswith("s_cbranch"):
                t = index.get(branch_target(a))
                out = []
                if a.op != "s_branch" and i + 1 < n:
                    out.append(i + 1)
                if t is not None:
                    out.append(t)
                    leaders.add(t)
                self.succ[i] = tuple(out)
                if i + 1 < n:
                    leaders.add(i + 1)
            elif a.op.startswith(("s_endpgm", "s_setpc", "s_swappc")):
                if i + 1 < n:
                    leaders.add(i + 1)
            elif i + 1 < n:
                self.succ[i] = (i + 1,)
        heads = sorted(leaders)
        self.block_of = [0] * n
        self.blocks = []                                    # (first, last) instruction index
        for b, h in enumerate(heads):
            last = (heads[b + 1] if b + 1 < len(heads) else n) - 1
            self.blocks.append((h, last))
            for i in range(h, last + 1):
                self.block_of[i] = b
        nb = len(self.blocks)
        bsucc = [[self.block_of[t] for t in self.succ[last]] for (_, last) in self.blocks]
        bpred = [[] for _ in range(nb)]
        for b, out in enumerate(bsucc):
            for t in out:
                bpred[t].append(b)
        # reverse post-order from the entry, then the iterative dominator algorithm (Cooper, Harvey, Kennedy)
        order, seen, stack = [], [False] * nb, [(0, 0)]
        seen[0] = True
        while stack:
            b, k = stack.pop()
            if k < len(bsucc[b]):
                stack.append((b, k + 1))
                t = bsucc[b][k]
                if not seen[t]:
                    seen[t] = True
                    stack.append((t, 0))
            else:
                order.append(b)
        rpo = order[::-1]
        num = {b: k for k, b in enumerate(rpo)}
        idom = {0: 0}
        changed = True
        while changed:
            changed = False
            for b in rpo[1:]:
                new = None
                for p_ in bpred[b]:
                    if p_ not in idom:
                        continue
                    if new is None:
                        new = p_
                        continue
                    x, y = p_, new
                    while x != y:
                        while num[x] > num[y]:
                            x = idom[x]
                        while num[y] > num[x]:
                            y = idom[y]
                    new = x
                if new is not None and idom.get(b) != new:
                    idom[b] = new
                    changed = True

        def dominates(a_, b_):
            while True:
                if a_ == b_:
                    return True
                if b_ == 0 or b_ not in idom:
                    return False
                b_ = idom[b_]

        # back edge u -> h (h dominates u) and the blocks of its natural loop
        self.loops = {}                                     # (last instruction of u, first of h) -> set of blocks
        for u in range(nb):
            for h in bsucc[u]:
                if u in idom and dominates(h, u):
                    body, work = {h, u}, [u]
                    while work:
                        x = work.pop()
                        if x == h:
                            continue
                        for p_ in bpred[x]:
                            if p_ not in body:
                                body.add(p_)
                                work.append(p_)
                    key = (self.blocks[u][1], self.blocks[h][0])
                    self.loops[key] = self.loops.get(key, set()) | body


def lint_kernel(name, ins):
    found = []
    flow = Flow(ins)
    succ = flow.succ

    def walk(i, budget, pipe=0):
        """(instruction index, wait states since instruction i) of everything within `budget` wait states behind i,
        along every path.  pipe = passes of MFMA i: a later MFMA of the same kind issues only once the matrix pipe has
        taken the last pass of the one before it, so by then instruction i is at least `pipe` wait states old."""
        best = {}
        stack = [(t, 0) for t in succ[i]]
        while stack:
            j, since = stack.pop()
            if pipe and ins[j].op == ins[i].op:
                since = max(since, pipe)
            if since >= budget or best.get(j, budget) <= since:
                continue
            best[j] = since
            for t in succ[j]:
                stack.append((t, since + ins[j].ws))
        return sorted(best.items())

    for i, a in enumerate(ins):
        # ---- (i) MFMA destination -> next use
        if a.is_mfma and a.ops and a.ops[0]:
            dst = a.ops[0][0]
            w, wmem = mfma_wait(a.op)
            for j, since in walk(i, max(w, wmem), 0 if STRICT else mfma_passes(a.op)):
                b = ins[j]
                hit = None
                if b.is_mfma:
                    # SrcC / destination identical to this destination: the accumulation chain, no software wait
                    for k, opnd in enumerate(b.ops):
                        for r in opnd:
                            if overlap(r, dst) and not (k in (0, 3) and r == dst):
                                hit = "MFMA reads it as %s" % ("SrcA/B" if k in (1, 2) else "a different SrcC tuple")
                    need = w
                else:
                    for opnd in b.ops:
                        for r in opnd:
                            if overlap(r, dst):
                                hit = "%s touches %s%d" % (b.op, r[0], r[1])
                    need = wmem if b.is_store else w
                if hit and since < need:
                    found.append("(i)   %s @%x: result %s[%d:%d] -- %s after %d of %d wait states (@%x)"
                                 % (a.op, a.addr, dst[0], dst[1], dst[2], hit, since, need, b.addr))
                    break
        # ---- (ii) VALU write -> MFMA read
        if a.is_valu and a.ops and a.ops[0] and not a.op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            dsts = list(a.ops[0])
            if a.op.startswith(("v_swap", "v_permlane")) and len(a.ops) > 1:
                dsts += a.ops[1]
            for j, since in walk(i, 2):
                b = ins[j]
                if b.is_mfma:
                    for opnd in b.ops[1:]:
                        for r in opnd:
                            if any(overlap(r, d) for d in dsts):
                                found.append("(ii)  %s @%x writes %s%d, %s @%x reads it after %d of 2 wait states"
                                             % (a.op, a.addr, r[0], r[1], b.op, b.addr, since))
    # ---- (iii) LDS-direct loads against the ds_reads of the next iteration (of the iteration `depth` later)
    depth = 1
    for pat, d in RING_DEPTH:
        if pat.search(name):
            depth = d
    for i, a in enumerate(ins):
        if not a.lds_direct:
            continue
        mine = flow.block_of[i]
        back = {e for e, body in flow.loops.items() if mine in body}
        if not back:
            continue                                        # not in a loop
        # state (instruction, crossed a back edge of a loop that holds the load) -> fewest vector-memory
        # instructions issued behind the load on any path there (the fewest is the worst case for a vmcnt(n))
        best, stack, bad = {}, [(t, 0, 0) for t in succ[i]], None
        while stack and bad is None:
            j, wrapped, later = stack.pop()
            if best.get((j, wrapped), 1 << 30) <= later:
                continue
            best[(j, wrapped)] = later
            b = ins[j]
            if b.op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", b.args)
                if m and int(m.group(1)) <= later:
                    continue                                # this path is covered
            if wrapped >= depth and (b.op.startswith("ds_read") or b.op.startswith("ds_load")):
                bad = b
                break
            if j == i and wrapped > depth:
                continue                                    # the whole iteration that reads the data has been walked
            nxt = later + (1 if b.vmem else 0)
            for t in succ[j]:
                stack.append((t, min(depth + 1, wrapped + (1 if (j, t) in back else 0)), nxt))
        if bad is not None:
            found.append("(iii) %s @%x (LDS-direct): %s @%x of the %s is reachable before an "
                         "s_waitcnt vmcnt that covers the load"
                         % (a.op, a.addr, bad.op, bad.addr,
                            "next loop iteration" if depth == 1 else "loop iteration %d later" % depth))
    return found


def demangle(name):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), shutil.which("c++filt")):
        if tool and os.path.exists(tool):
            return subprocess.run([tool, name], capture_output=True, text=True).stdout.strip()
    return name


def lint_file(path, verbose=False):
    kernels = disassemble(path)
    total, n_mfma, n_lds = [], 0, 0
    for name, ins in sorted(kernels.items()):
        n_mfma += sum(1 for a in ins if a.is_mfma)
        n_lds += sum(1 for a in ins if a.lds_direct)
        f = lint_kernel(name, ins)
        if f:
            total.append((name, f))
    print("%s: %d kernels, %d MFMAs, %d LDS-direct loads, %d kernels with findings"
          % (os.path.relpath(path, ROOT) if path.startswith(ROOT) else path, len(kernels), n_mfma, n_lds, len(total)))
    for name, f in total:
        short = demangle(name)
        print("  %s" % (short or name)[:150])
        for line in f[: (len(f) if verbose else 6)]:
            print("      " + line)
        if len(f) > 6 and not verbose:
            print("      ... %d more" % (len(f) - 6))
    return total


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    verbose = "-v" in sys.argv[1:]
    global STRICT
    STRICT = "--strict" in sys.argv[1:]
    paths = args or [os.path.join(ROOT, "speaker-recognition_amd", "libskm.so")]
    bad = 0
    for p in paths:
        bad += len(lint_file(p, verbose))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
