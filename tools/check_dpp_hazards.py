#!/usr/bin/env python3
"""Static check of the gfx9 DPP read-after-VALU-write hazard in a compiled kernel.

hipcc pads the hazards of the instructions it emits but nothing inside an
`asm` string, and our fused `v_fmac_f64_dpp ... row_newbcast` blocks live in
asm strings.  The hardware needs 2 wait states between a VALU write of a VGPR
and a DPP instruction reading it as its (lane-permuted) src0, and 5 between a
VALU write of EXEC and any DPP instruction.  This script walks the `.s` of the
device code (hipcc -save-temps) and reports every DPP read whose source
register was written by a VALU instruction fewer than 2 wait states earlier on
ANY path (fall-through or branch), and every v_cmpx within 5 wait states of a
DPP instruction.  It also checks the gfx940+ TRANS forwarding hazard that an asm block with a
transcendental in it has to respect by itself: one wait state between a TRANS instruction (v_rcp,
v_rsq, v_sqrt, v_exp, v_log, v_sin, v_cos) and a non-TRANS VALU instruction that reads its result.

usage: check_dpp_hazards.py file.s [kernel-name-substring ...]; exit code 1 on a hazard.
"""
import re
import sys

VALU_PREFIX = ("v_",)
NOT_VALU_WRITERS = ("v_cmp_", "v_cmpx_", "v_readlane", "v_readfirstlane", "v_nop")


def regs(token):
    token = token.strip().lstrip("-|").rstrip("|,")
    m = re.match(r"^v\[(\d+):(\d+)\]$", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", token)
    if m:
        return {int(m.group(1))}
    return set()


def parse_kernel(lines):
    """-> list of items: ('label', name) | ('ins', mnemonic, operands[list], text)"""
    items = []
    for raw in lines:
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", line)
        if m and not line.startswith("\t"):
            items.append(("label", m.group(1)))
            continue
        s = line.strip()
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        mnem = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        items.append(("ins", mnem, ops, s))
    return items


def wait_states(item):
    if item[1] == "s_nop":
        return int(item[2][0], 0) + 1
    return 1


def valu_written(item):
    mnem, ops = item[1], item[2]
    if not mnem.startswith(VALU_PREFIX) or mnem.startswith(NOT_VALU_WRITERS) or not ops:
        return set()
    return regs(ops[0])


def dpp_source(item):
    mnem, ops, text = item[1], item[2], item[3]
    if "_dpp" not in mnem and "row_newbcast" not in text and "quad_perm" not in text \
            and "row_shr" not in text and "row_shl" not in text and "row_ror" not in text \
            and "row_bcast" not in text and "row_mirror" not in text and "wave_" not in text:
        return None
    if len(ops) < 2:
        return set()
    src0 = ops[1].split()[0]
    return regs(src0)


TRANS_PREFIX = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def valu_read(item):
    mnem, ops = item[1], item[2]
    if not mnem.startswith(VALU_PREFIX):
        return set()
    out = set()
    for o in ops[1:]:
        out |= regs(o.split()[0]) if o.split() else set()
    if "fmac" in mnem or "_mac_" in mnem:  # accumulator is read too
        out |= regs(ops[0])
    return out


def check(items, name):
    # control-flow: index of labels, predecessors by branch
    label_at = {it[1]: i for i, it in enumerate(items) if it[0] == "label"}
    branch_preds = {}
    for i, it in enumerate(items):
        if it[0] == "ins" and (it[1].startswith("s_cbranch") or it[1] == "s_branch"):
            tgt = it[2][0] if it[2] else None
            if tgt in label_at:
                branch_preds.setdefault(label_at[tgt], []).append(i)

    def preceding(i, budget):
        """Yield (instruction, wait states between it and instruction i) along every
        backward path (fall-through and branches), up to `budget` wait states."""
        stack = [(i - 1, 0)]
        seen = set()
        while stack:
            j, ws = stack.pop()
            while j >= 0 and ws < budget and (j, ws) not in seen:
                seen.add((j, ws))
                it = items[j]
                if it[0] == "label":
                    for b in branch_preds.get(j, []):
                        stack.append((b, ws))
                    k = j - 1
                    while k >= 0 and items[k][0] == "label":
                        k -= 1
                    if k >= 0 and items[k][1] in ("s_branch", "s_endpgm", "s_setpc_b64"):
                        break  # no fall-through into this label
                    j -= 1
                    continue
                yield it, ws
                ws += wait_states(it)
                j -= 1

    problems = []
    for i, it in enumerate(items):
        if it[0] != "ins":
            continue
        if it[1].startswith(VALU_PREFIX) and not it[1].startswith(TRANS_PREFIX):
            reads = valu_read(it)
            for prev, ws in preceding(i, 1):
                if prev[1].startswith(TRANS_PREFIX) and ws < 1 and (valu_written(prev) & reads):
                    problems.append((name, prev[3], it[3], ws, "TRANS result -> VALU read needs 1 wait state"))
        src = dpp_source(it)
        if src is None:
            continue
        for prev, ws in preceding(i, 5):
            if ws < 2 and (valu_written(prev) & src):
                problems.append((name, prev[3], it[3], ws, "VALU write -> DPP read needs 2 wait states"))
            if prev[1].startswith("v_cmpx") and ws < 5:
                problems.append((name, prev[3], it[3], ws, "VALU EXEC write -> DPP needs 5 wait states"))
    return problems


VMEM_PREFIX = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic",
               "flat_load", "flat_store", "scratch_load", "scratch_store")


def is_vmem(item):
    return item[0] == "ins" and item[1].startswith(VMEM_PREFIX)


def is_lds_dma(item):
    return item[0] == "ins" and (item[1].startswith("global_load_lds") or
                                 (item[1].startswith("buffer_load") and " lds" in (" " + item[3])))


def check_counted_waits(items, name):
    """The staged kernels conclude from a COUNTED wait that an LDS-DMA group has landed
    (chain_qw16.hpp): `s_waitcnt vmcnt(k)`, k > 0, lets the k youngest vector-memory operations stay in
    flight, so that is only right if, on EVERY path into the wait, those k youngest operations are not
    the DMA whose image the code behind the wait reads.  Two patterns are legitimate there:
      (stores)  the k youngest operations are all non-DMA (the spill stores of the previous stage);
      (group)   youngest-first, the operations are [non-DMA]* then DMA only: the wait then reaches at most
                into the NEWEST DMA group (a longer wait than needed is harmless), never across older
                non-DMA operations into a group before it.
    Flagged: a path with fewer than k vector-memory operations before the kernel entry is fine (everything
    has landed); a path whose k youngest operations run DMA -> non-DMA -> DMA (the wait would leave part of
    an OLDER group in flight), and any LDS-DMA kernel that spills registers (scratch traffic would shift
    every count).  Not proven by this check: that a wait meant to cover only stores has at least k stores
    in front of it on every path, and reaches between two groups that are adjacent on a path."""
    label_at = {it[1]: i for i, it in enumerate(items) if it[0] == "label"}
    branch_preds = {}
    for i, it in enumerate(items):
        if it[0] == "ins" and (it[1].startswith("s_cbranch") or it[1] == "s_branch"):
            tgt = it[2][0] if it[2] else None
            if tgt in label_at:
                branch_preds.setdefault(label_at[tgt], []).append(i)
    problems = []
    if not any(is_lds_dma(it) for it in items):
        return problems
    for it in items:
        if it[0] == "ins" and it[1].startswith("scratch_"):
            problems.append((name, it[3], "", 0, "LDS-DMA kernel with scratch traffic: counted vmcnt waits are unsafe"))
            break
    for i, it in enumerate(items):
        if it[0] != "ins" or it[1] != "s_waitcnt":
            continue
        m = re.search(r"vmcnt\((\d+)\)", it[3])
        if not m or int(m.group(1)) == 0:
            continue
        k = int(m.group(1))
        # depth-first over backward paths; state = (index, ops seen, phase) with phase 0: only non-DMA seen,
        # 1: inside the DMA run, 2: non-DMA after a DMA run (a later DMA is then the violation)
        stack = [(i - 1, 0, 0)]
        seen = set()
        bad = None
        while stack and bad is None:
            j, count, phase = stack.pop()
            while j >= 0 and count < k:
                if (j, count, phase) in seen:
                    break
                seen.add((j, count, phase))
                cur = items[j]
                if cur[0] == "label":
                    for b in branch_preds.get(j, []):
                        stack.append((b, count, phase))
                    q = j - 1
                    while q >= 0 and items[q][0] == "label":
                        q -= 1
                    if q >= 0 and items[q][1] in ("s_branch", "s_endpgm", "s_setpc_b64"):
                        break
                    j -= 1
                    continue
                if is_vmem(cur):
                    count += 1
                    dma = is_lds_dma(cur)
                    if phase == 0 and dma:
                        phase = 1
                    elif phase == 1 and not dma:
                        phase = 2
                    elif phase == 2 and dma:
                        bad = cur[3]
                        break
                j -= 1
        if bad is not None:
            problems.append((name, bad, it[3], k, "counted vmcnt reaches across non-DMA operations into an older LDS-DMA group"))
    return problems


def main():
    path = sys.argv[1]
    wanted = sys.argv[2:]
    lines = open(path).read().split("\n")
    kernels = {}
    cur = None
    for ln in lines:
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*;?\s*@", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None:
            kernels[cur].append(ln)
            if ln.strip().startswith("s_endpgm"):
                cur = None
    total = 0
    checked = 0
    for name, body in kernels.items():
        if wanted and not any(w in name for w in wanted):
            continue
        items = parse_kernel(body)
        ndpp = sum(1 for it in items if it[0] == "ins" and dpp_source(it) is not None)
        probs = check(items, name) + check_counted_waits(items, name)
        checked += 1
        print(f"{name}: {ndpp} DPP instructions, {len(probs)} hazard(s)")
        for p in probs[:20]:
            print(f"   [{p[4]}; {p[3]} wait state(s)]\n      writer: {p[1]}\n      reader: {p[2]}")
        total += len(probs)
    if checked == 0:
        print("no kernel matched")
        return 2
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
