#!/usr/bin/env python3
"""Static instruction mix of one instantiation of ldpc_decode2_kernel from its device assembly, and the issue cost of that mix with the
per-class cycles measured by tools/ubench/valu_rate (json lines). A static count is not the dynamic mix (loops, layer kinds), but the
vector stream of this kernel is the same few op classes everywhere.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip --cuda-device-only -S sdr_receiver_dvb_t2_amd/csrc/ldpc_kernel2.hip -o /tmp/ldpc2.s
    python tools/ldpc_isa_mix.py /tmp/ldpc2.s ILi12ELi12ELi4E [gpurun_out/valu_rate.txt]
"""
import collections
import json
import re
import sys

asm, inst = sys.argv[1], sys.argv[2]
rates = {}
if len(sys.argv) > 3:
    for ln in open(sys.argv[3]):
        if ln.startswith("json "):
            d = json.loads(ln[5:])
            rates[d["op"]] = d
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if "ldpc_decode2_kernel" + inst in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\S+:", l))
body = []
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    body.append(l)
ops = collections.Counter()
for l in body:
    m = re.match(r"^\s+([vsd][a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|flat_[a-z0-9_]+)\b", l)
    if m:
        ops[m.group(1)] += 1
valu = {k: v for k, v in ops.items() if k.startswith("v_")}
nv = sum(valu.values())
print("instructions: %d total, %d VALU, %d SALU, %d LDS, %d global/buffer" % (
    sum(ops.values()), nv, sum(v for k, v in ops.items() if k.startswith("s_")), sum(v for k, v in ops.items() if k.startswith("ds_")),
    sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "flat_")))))
# class of every VALU opcode -> the ubench op that stands for it
CLASS = [
    (r"v_pk_(min|max)_[iu]16", "pk_min_i16"), (r"v_pk_(add|sub)_[iu]16", "pk_sub_i16_clamp"), (r"v_pk_mad_[iu]16|v_pk_mul_lo_u16", "pk_mad_i16"),
    (r"v_pk_(ashr|lshr|lshl)rev_[ib]16", "pk_ashr_i16"), (r"v_perm_b32", "perm_b32"), (r"v_.*_sdwa", "sub_u16_sdwa"), (r"v_.*_dpp|v_mov_b32_dpp", "mov_dpp"),
    (r"v_med3_", "med3_i32"), (r"v_(min|max)_[iu](32|16)", "min_i32"), (r"v_bfe_|v_bfi_|v_alignbit|v_lshl_or|v_and_or|v_or3|v_xad|v_lshl_add|v_add_lshl|v_add3|v_mad_", "add3_u32"),
    (r"v_cmp|v_cmpx", "cmp_only"), (r"v_cndmask", "cmp_cnd"), (r"v_(add|sub|subrev)_(u32|co_u32|nc_u32|u16|i32)", "add_u32"),
    (r"v_(lshl|lshr|ashr)rev_[bi]32", "lshlrev_b32"), (r"v_(and|or|xor|not)_b32|v_mov_b32|v_xnor", "xor_b32"),
    (r"v_readlane|v_readfirstlane|v_writelane", "add_u32"),
]
by = collections.Counter()
other = collections.Counter()
for k, v in valu.items():
    for pat, cls in CLASS:
        if re.match(pat, k):
            by[cls] += v
            break
    else:
        other[k] += v
print("VALU by class (static share)%s:" % (", cycles per wave instruction at 3 waves / SIMD from the ubench" if rates else ""))
tot = 0.0
for cls, v in by.most_common():
    c = rates.get(cls, {}).get("cycles_3w")
    tot += v * (c if c else 4.0)
    print("  %-18s %6d  %5.1f %%  %s" % (cls, v, 100.0 * v / nv, ("%.2f" % c) if c else ""))
if other:
    print("  unclassified (taken at 4 cycles): %s" % dict(other))
    tot += 4.0 * sum(other.values())
print("mix-weighted cycles per VALU wave instruction: %.2f" % (tot / nv))
out = {"valu_mix_cycles": round(tot / nv, 3), "valu_cycles_per_wave_inst": {k: rates[k]["cycles_3w"] for k in by if k in rates},
       "valu_static_share": {k: round(v / nv, 3) for k, v in by.most_common()}}
print(json.dumps(out))
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
