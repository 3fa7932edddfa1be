#!/usr/bin/env python3
"""The instruction stream ONE wavefront issues for a PLAIN layer of ldpc_decode2_kernel<12,12,4> (64800 r = 3/4: 25 of its 45 layers),
from the device assembly, every vector instruction with its issue class and the cycles that class costs at three wavefronts per
SIMD (tools/ubench/valu_rate, profiles/r03_valu_rate.txt), and the floor the list implies for a layer.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip --cuda-device-only -S sdr_receiver_dvb_t2_amd/csrc/ldpc_kernel2.hip -o /tmp/ldpc2.s
    python tools/ldpc_plain_layer.py /tmp/ldpc2.s profiles/r03_valu_rate.txt > profiles/r05_ldpc_plain_layer_isa.txt

The path is found by its shape, not by label numbers: the innermost layer loop's header, the load phase behind it (the block with
the seven ds_read_u16), and the block that begins with the chain of v_pk_min_u16 / v_pk_max_u16 over all seven slots and stores
seven ds_write_b16 (the PLAIN write phase). Branches of the other layer kinds are not followed."""
import collections
import json
import re
import sys

asm, rate_file = sys.argv[1], sys.argv[2]
rates = {}
for ln in open(rate_file):
    if ln.startswith("json "):
        d = json.loads(ln[5:])
        rates[d["op"]] = d["cycles_3w"]
CLASS = [
    (r"v_pk_(min|max)_[iu]16", "pk_min_i16"), (r"v_pk_(add|sub)_[iu]16", "pk_sub_i16_clamp"), (r"v_pk_mad_[iu]16|v_pk_mul_lo_u16", "pk_mad_i16"),
    (r"v_pk_(ashr|lshr|lshl)rev_[ib]16", "pk_ashr_i16"), (r"v_perm_b32", "perm_b32"), (r"v_.*_sdwa", "sub_u16_sdwa"), (r"v_.*_dpp", "mov_dpp"),
    (r"v_med3_", "med3_i32"), (r"v_(min|max)_[iu](32|16)", "min_i32"),
    (r"v_bfe_|v_bfi_|v_alignbit|v_lshl_or|v_and_or|v_or3|v_xad|v_lshl_add|v_add_lshl|v_add3|v_mad_|v_bitop3", "add3_u32"),
    (r"v_cmp|v_cmpx", "cmp_only"), (r"v_cndmask", "cmp_cnd"), (r"v_(add|sub|subrev)_(u32|co_u32|nc_u32|u16|i32)", "add_u32"),
    (r"v_(lshl|lshr|ashr)rev_[bi]32", "lshlrev_b32"), (r"v_(and|or|xor|not)_b32|v_mov_b32|v_mov_b64|v_xnor", "xor_b32"),
    (r"v_readlane|v_readfirstlane|v_writelane", "add_u32"),
]


def cls(op):
    for pat, c in CLASS:
        if re.match(pat, op):
            return c
    return None


lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*ldpc_decode2_kernelILi12ELi12ELi4E\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
# basic blocks
blocks, cur, name = [], [], "entry"
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
    elif re.match(r"^\s+[a-z]", l) and not l.strip().startswith(";"):
        cur.append(l.strip().split(";")[0].strip())
        if cur[-1].startswith(("s_cbranch", "s_branch")):                 # a block ends at its branch
            blocks.append((name, cur)); cur, name = [], name + "+"
blocks.append((name, cur))


def ops(b):
    return [x.split()[0] for x in b]


# the load phase: seven ds_read_u16; the write phase: seven ds_write_b16 and no ds_read_u16
load_i = next(i for i, (_, b) in enumerate(blocks) if ops(b).count("ds_read_u16") == 7)
write_i = next(i for i, (_, b) in enumerate(blocks) if ops(b).count("ds_write_b16") >= 7 and ops(b).count("ds_read_u16") == 0 and
               sum(o.startswith(("v_pk_min", "v_pk_max")) for o in ops(b)) >= 20 and not any(o.endswith("_dpp") and False for o in ops(b)))
# the head: from the loop header (the block with the four v_readlane_b32 of the layer table) up to the load phase
head_i = max(i for i, (_, b) in enumerate(blocks[:load_i]) if ops(b).count("v_readlane_b32") >= 4)
# (the write phase's block is entered through a short block that ends in the branch around it for wavefronts without a node)
w_parts = [x for _, b in blocks[write_i - 1:write_i + 1] for x in b] if ops(blocks[write_i - 1][1])[-1:] == ["s_cbranch_vccnz"] else blocks[write_i][1]
parts = [("head: layer table out of registers, the lane's parity address, record prefetch + delayed record store (the rare branches -- layer 0's wrap, level-scheduled layers -- listed too)",
          [x for _, b in blocks[head_i:load_i] for x in b]),
         ("load: 7 slots -- address, LLR pair, old message, in = sat(L - msg), |in|; next layer's entries prefetched", blocks[load_i][1]),
         ("update: two smallest |in| over the node (7 slots x 2 lanes), sign product, 7 new LLR pairs stored, 7 new messages packed", w_parts)]
tot = collections.Counter()
cyc_tot = 0.0
print("# one wavefront, one PLAIN layer of ldpc_decode2_kernel<12,12,4> (tools/ldpc_plain_layer.py); cycles = issue cost of the instruction's")
print("# class at 3 wavefronts per SIMD (profiles/r03_valu_rate.txt); scalar / LDS / memory instructions issue beside the vector pipe")
for title, ins in parts:
    n_v = n_s = n_l = n_g = 0
    cyc = 0.0
    print("\n## %s" % title)
    for x in ins:
        op = x.split()[0]
        if op.startswith("v_"):
            c = cls(op)
            k = rates.get(c, 4.37)
            n_v += 1; cyc += k; tot[c or "other"] += 1
            print("  %-58s %-18s %4.2f" % (x[:58], c or "other", k))
        elif op.startswith("ds_"):
            n_l += 1; print("  %-58s LDS" % x[:58])
        elif op.startswith(("global_", "buffer_", "flat_")):
            n_g += 1; print("  %-58s memory" % x[:58])
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
            n_s += 1
    print("  -> %d vector (%.0f cycles), %d LDS, %d memory, %d scalar" % (n_v, cyc, n_l, n_g, n_s))
    cyc_tot += cyc
    tot["_v"] += n_v
print("\n## sum")
print("vector instructions per wavefront and layer: %d (~10 of the head's are in branches a PLAIN layer behind layer 0 does not take)" % tot["_v"])
print("by class: " + ", ".join("%s %d" % (k, v) for k, v in tot.most_common() if k != "_v"))
print("issue cost: %.0f cycles per wavefront, x 3 wavefronts per SIMD = %.0f cycles per layer if nothing else ever waits" % (cyc_tot, 3 * cyc_tot))
print("(measured, tools/ldpc_phase_profile.py: a PLAIN layer takes 2050 - 2250 cycles of s_memtime: the layer IS its vector issue.)")
