"""Static hazard lint of the gfx950 machine code of the library (toolchain canary; VERDICT r05 task 2).  Two rules, no GPU needed.

Rule 1 -- in-flight vector-memory destinations.  On gfx9-family hardware vector-memory operations complete in issue order and
`s_waitcnt vmcnt(N)` waits until at most N are outstanding; walking a kernel's instructions along its control-flow graph with the queue
of outstanding vector-memory operations (destination registers per entry; fixed point over basic blocks, queues merged from the most
recent entry backwards), an instruction that WRITES (other than a later load: loads return in order) or READS a register some
still-outstanding load will write is a missing wait count.  (This is what round 5 believed the default-scheduler fault to be; it is
not -- that build is clean under rule 1 and still faults with a forced full wait after every instruction.)

Rule 2 -- register-allocator traffic in front of a join block's exec restore (`lint_exec_prologue`).  THE cause of the fault of
profiles/r05_experiments.txt item 24, found in round 6 with the ROCm debug agent (profiles/r06_experiments.txt item 7): a divergent
`if (tid == 0) { poll a flag }` is lowered to `s_and_saveexec_b64 ; s_cbranch_execz JOIN ; ... ; JOIN: s_or_b64 exec, exec, sN`, and under
the DEFAULT machine scheduler of clang 22 / ROCm 7.2.0 the register allocator parks a whole-wave VGPR in an AGPR with its
`v_accvgpr_write_b32` placed between the JOIN label and the `s_or_b64` -- written under the then-branch's one-lane mask (or under exec
= 0), read back later for all 64 lanes: garbage lane offsets, GPU memory fault.  The rule reports AGPR copies and scratch spills /
reloads at the target of a forward `s_cbranch_execz` in front of the exec restore.  It flags the faulty build (exactly that
instruction) and passes the shipped one; the source no longer has that divergent region (tb_stepx_kernels.hip: the poll is
wave-uniform), so the default-scheduler build of the current sources is clean too and runs (profiles/r06_experiments.txt item 7).

`__graft_entry__.build()` runs both rules over every object it has just built and refuses to ship a library with a finding; planted
cases for both rules: tools/microtests/waw_case.s (tests/test_abi_and_host.py).

usage: python tools/isa_waw_lint.py [objects or .so ...]      (default: trafficbots_amd/csrc/build/*.o)
"""
from __future__ import annotations

import glob
import os
import re
import subprocess
import sys
import tempfile
from typing import Dict, List, Optional, Set, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("TB_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
MAX_Q = 96  # queue entries kept per state (vmcnt is a 6-bit counter: 63 outstanding at most)

_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_VMEM_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load|tbuffer_load|global_atomic|buffer_atomic|flat_atomic|image_load|image_sample)")
_VMEM_STORE = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store|tbuffer_store|image_store)")
_BRANCH = re.compile(r"^s_c?branch")


def regs(text: str) -> Set[Tuple[str, int]]:
    out: Set[Tuple[str, int]] = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


class Ins:
    __slots__ = ("addr", "op", "dst", "src", "kind", "vmcnt", "target", "text")

    def __init__(self, addr: int, text: str) -> None:
        self.addr, self.text = addr, text
        parts = text.split(None, 1)
        self.op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        self.kind, self.vmcnt, self.target = "other", None, None
        self.dst: Set[Tuple[str, int]] = set()
        self.src: Set[Tuple[str, int]] = set()
        if self.op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", text)
            self.kind = "wait"
            self.vmcnt = int(m.group(1)) if m else None  # None: the vmcnt field is at its maximum (no wait on it)
            return
        if _BRANCH.match(self.op) or self.op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            self.kind = "branch"
            return
        if _VMEM_LOAD.match(self.op):
            self.kind = "vmem"
            to_lds = " lds" in (" " + text)
            returns = not self.op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")) or " glc" in text or " sc0" in text
            if ops and not to_lds and returns:
                self.dst = regs(ops[0])
                self.src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            else:
                self.src = set().union(*[regs(o) for o in ops]) if ops else set()
            return
        if _VMEM_STORE.match(self.op):
            self.kind = "vmem"
            self.src = set().union(*[regs(o) for o in ops]) if ops else set()
            return
        # everything else: the first operand is the destination when it is a vector register (VALU, MFMA, ds_read, v_readlane -> SGPR ...);
        # instructions without a vector destination (ds_write, v_cmp to vcc / SGPRs, s_*) only read
        no_dst = self.op.startswith(("ds_write", "ds_store", "v_cmp_", "v_cmpx_", "s_", "ds_append", "ds_gws", "buffer_wbl2", "buffer_inv"))
        if ops and not no_dst:
            self.dst = regs(ops[0])
            self.src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if self.op.startswith(("v_mac_", "v_fmac_", "v_dot", "v_pk_fmac")) or "mfma" in self.op and len(ops) >= 4:
                self.src |= regs(ops[-1]) if "mfma" in self.op else self.dst  # accumulating forms read their destination
        else:
            self.src = set().union(*[regs(o) for o in ops]) if ops else set()


def device_code(path: str, workdir: str) -> List[str]:
    """Extract the gfx950 code object(s) of a host object / shared library and return their disassembly lines."""
    fat = os.path.join(workdir, os.path.basename(path) + ".fat")
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path, os.path.join(workdir, "unused.o")],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return []
    lines: List[str] = []
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    for n, s in enumerate(starts):  # a linked library holds one bundle per translation unit, back to back
        e = starts[n + 1] if n + 1 < len(starts) else len(data)
        one = f"{fat}.{n}"
        open(one, "wb").write(data[s:e])
        co = f"{one}.co"
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={one}", f"--targets={TARGET}", f"--output={co}"],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co):
            continue
        d = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True)
        lines += d.stdout.splitlines()
    return lines


def parse_asm_text(lines: List[str]) -> Dict[str, List[Ins]]:
    """Kernels of a `clang -S` assembly listing (labels instead of addresses): label -> pseudo address = instruction index."""
    fn: Dict[str, List[Tuple[str, Optional[str]]]] = {}
    cur, pending = None, []
    labels: Dict[str, Dict[str, int]] = {}
    for ln in lines:
        t = ln.split(";")[0].split("//")[0].rstrip()
        if not t.strip():
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):\s*$", t)
        if m:
            name = m.group(1)
            if not name.startswith(".L"):
                cur = name
                fn[cur], labels[cur] = [], {}
            elif cur is not None:
                labels[cur][name] = len(fn[cur])
            continue
        if cur is None or t.lstrip().startswith("."):
            continue
        fn[cur].append(t.strip())
    out: Dict[str, List[Ins]] = {}
    for name, body in fn.items():
        if not body:
            continue
        ins = []
        for i, t in enumerate(body):
            x = Ins(i, t)
            if x.kind == "branch":
                m = re.search(r"(\.L[\w.$]+)", t)
                if m and m.group(1) in labels[name]:
                    x.target = labels[name][m.group(1)]
            ins.append(x)
        out[name] = ins
    return out


def parse_objdump(lines: List[str]) -> Dict[str, List[Ins]]:
    out: Dict[str, List[Ins]] = {}
    cur = None
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln.strip())
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m and cur is not None:
            x = Ins(int(m.group(2), 16), m.group(1).strip())
            if x.kind == "branch":
                t = re.search(r"<.+\+0x([0-9a-fA-F]+)>", ln)
                base = out[cur][0].addr if out[cur] else x.addr
                if t:
                    x.target = base + int(t.group(1), 16)
                elif re.search(r"<[^+>]+>\s*$", ln):
                    x.target = base
            out[cur].append(x)
    return {k: v for k, v in out.items() if v}


def lint_kernel(ins: List[Ins]) -> List[str]:
    """Fixed point over basic blocks; state = tuple of frozensets (outstanding vector-memory destinations, oldest first)."""
    index = {x.addr: i for i, x in enumerate(ins)}
    leaders = {0}
    for i, x in enumerate(ins):
        if x.kind == "branch":
            if x.target is not None and x.target in index:
                leaders.add(index[x.target])
            if i + 1 < len(ins):
                leaders.add(i + 1)
    order = sorted(leaders)
    block_end = {b: (order[n + 1] if n + 1 < len(order) else len(ins)) for n, b in enumerate(order)}
    state_in: Dict[int, Optional[Tuple[frozenset, ...]]] = {b: None for b in order}
    state_in[0] = ()
    hazards: Dict[int, str] = {}

    def merge(a, b):
        if a is None:
            return b
        n = max(len(a), len(b))
        pa, pb = (frozenset(),) * (n - len(a)) + a, (frozenset(),) * (n - len(b)) + b
        return tuple(x | y for x, y in zip(pa, pb))

    work = [0]
    rounds = 0
    while work and rounds < 20000:
        rounds += 1
        b = work.pop()
        q = list(state_in[b])
        i = b
        succ = []
        while i < block_end[b]:
            x = ins[i]
            if x.kind == "wait":
                if x.vmcnt is not None:
                    q = q[len(q) - x.vmcnt:] if x.vmcnt < len(q) else q
                    if x.vmcnt == 0:
                        q = []
            else:
                pend = frozenset().union(*q) if q else frozenset()
                if pend:
                    # (a LOAD into a register an earlier, still outstanding load will write is not a hazard: vector-memory loads return
                    # in order, the later data lands last -- the compiler re-uses dead load destinations that way)
                    w = (x.dst & pend) if x.kind != "vmem" else frozenset()
                    r = x.src & pend
                    if w or r:
                        what = ("WRITES " + _fmt(w) if w else "") + (" READS " + _fmt(r) if r else "")
                        hazards.setdefault(x.addr, f"  @{x.addr:#x}: `{x.text}`{what} -- destination of a vector-memory load still outstanding "
                                                   f"({len(q)} in flight)")
                if x.kind == "vmem":
                    q.append(frozenset(x.dst))
                    if len(q) > 63:  # the counter saturates: the hardware stalls the issue until an entry retires
                        q = q[-63:]
            if x.kind == "branch":
                if x.target is not None and x.target in index:
                    succ.append(index[x.target])
                if x.op not in ("s_branch", "s_endpgm", "s_setpc_b64") and i + 1 < len(ins):
                    succ.append(i + 1)
                break
            i += 1
        else:
            if block_end[b] < len(ins):
                succ.append(block_end[b])
        out = tuple(q[-MAX_Q:])
        for s_ in succ:
            if s_ not in state_in:
                continue
            m = merge(state_in[s_], out)
            if m != state_in[s_]:
                state_in[s_] = m
                work.append(s_)
    return [hazards[a] for a in sorted(hazards)]


_EXEC_RESTORE = re.compile(r"^s_(or|mov|xor|andn2|and)_b64\s+exec\b")


def lint_exec_prologue(ins: List[Ins]) -> List[str]:
    """Second rule (round 6, found with the ROCm debug agent on the default-scheduler build: profiles/r06_experiments.txt item 7).
    A divergent `if` is lowered to `s_and_saveexec_b64 sN, vcc ; s_cbranch_execz JOIN ; ...then... ; JOIN: s_or_b64 exec, exec, sN`.
    The join block starts with the exec restore; a VECTOR instruction between the JOIN label and that restore runs under the
    then-branch's mask (or under exec = 0 when the branch was taken) although it belongs to the code after the `if`.  That is what the
    faulty build does: the register allocator's copy `v_accvgpr_write_b32 a73, v194` (a whole-wave value parked in an AGPR across the
    `if (tid == 0)` poll of the GRU helper flag) sits in front of `s_or_b64 exec, exec, s[4:5]` -- it writes lane 0 only (or no lane),
    the later `v_accvgpr_read_b32` hands 63 lanes of garbage to the weight-fragment address computation: GPU memory fault.
    Reported: register-allocator traffic in that position -- AGPR copies (`v_accvgpr_write / _read`) and scratch spills / reloads -- at
    the target of a FORWARD `s_cbranch_execz`.  (Ordinary vector instructions there are not reported: the skip branch of a divergent
    `if` may legally land inside the tail of the then-block, whose instructions are no-ops under exec = 0; and loop back-edges restore
    their masks differently.)"""
    index = {x.addr: i for i, x in enumerate(ins)}
    out: Dict[int, str] = {}
    for x in ins:
        if x.kind != "branch" or not x.op.startswith("s_cbranch_execz") or x.target not in index or x.target <= x.addr:
            continue
        j = index[x.target]
        pending = []
        while j < len(ins):
            y = ins[j]
            if _EXEC_RESTORE.match(y.text):
                for z in pending:
                    out.setdefault(z.addr, f"  @{z.addr:#x}: `{z.text}` sits in a join block IN FRONT OF the exec restore `{y.text}` "
                                           f"(target of `{x.text.split()[0]}` @{x.addr:#x}): it runs under the divergent branch's mask")
                break
            if y.kind == "branch" or y.op in ("s_barrier", "s_endpgm") or "exec" in y.text:
                break
            if y.op.startswith(("v_accvgpr_write", "v_accvgpr_read", "scratch_store", "scratch_load")):
                pending.append(y)
            j += 1
    return [out[a] for a in sorted(out)]


def _fmt(rs) -> str:
    return ",".join(f"{k}{i}" for k, i in sorted(rs))


def lint_paths(paths: List[str], only: Optional[str] = None) -> Tuple[int, int, List[str]]:
    n_kernel, report = 0, []
    with tempfile.TemporaryDirectory() as wd:
        for path in paths:
            if path.endswith(".s"):
                kernels = parse_asm_text(open(path).read().splitlines())
            else:
                kernels = parse_objdump(device_code(path, wd))
            for name, ins in kernels.items():
                if only and only not in name:
                    continue
                n_kernel += 1
                hz = lint_kernel(ins) + lint_exec_prologue(ins)
                if hz:
                    report.append(f"{os.path.basename(path)}: {name[:100]}: {len(hz)} hazard(s)")
                    report += hz[:8]
    return n_kernel, sum(1 for r in report if not r.startswith("  ")), report


def main() -> int:
    paths = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "trafficbots_amd", "csrc", "build", "*.o")))
    n_kernel, n_bad, report = lint_paths(paths)
    print(f"isa_waw_lint: {n_kernel} kernels in {len(paths)} file(s), {n_bad} with a hazard (in-flight vector-memory destination touched / "
          f"vector instruction in front of a join block's exec restore)")
    for r in report:
        print(r)
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
