#!/usr/bin/env python3
"""Writes multispectral-object-detection_amd/csrc/probes/bottleneck_asm.inc: the hand-scheduled 3x3 loop of bottleneck128a_kernel
(csrc/probes/bottleneck_asm.hip - probe build only: measured slower than the shipped kernel, profiles/r06_bottleneck128_asm.md) as inline-asm text, one string per (operand type, wave group).

The kernel: one 8-wave workgroup per 16 x 16-pixel tile of a 128-channel Bottleneck; the hidden tensor t = SiLU(W1 x + b1) of the
tile's 18 x 18 halo patch sits in LDS (two 64-channel planes of 128-byte pixel rows, 16-byte slot s of pixel q = k-granule
s ^ (q & 7)); the 3x3 conv is NINE K steps (one per tap, 128 channels = four 32-wide sub-steps), its weights streaming through
two 32-KiB LDS buffers by LDS-DMA, its A fragments SHIFTED reads of the patch.  Wave (wm, wn) = (wave >> 1, wave & 1) owns tile
rows 4 wm .. 4 wm + 3 (four 16-pixel m-tiles) x channels 64 wn .. + 63 (four n-tiles): 64 MFMAs per K step.

Register map of the asm block (per lane):
  v[128:191]  accumulators, acc[i][j] = v[128 + 4*(4i+j) : +3] - ARCHITECTURAL registers: a kernel that touches the accumulator file is held to an
              even 128 / 128 split of its 256 registers by hipcc (ROCm 7.2), and the W1 stage / epilogue around the loop need more than 128;
              the block hands them out through 32 64-bit output operands (o0..o31 = acc registers 2t, 2t + 1) at its end
  v[40:55]    fa set 0 (m-tiles 0..3)   v[56:71]  fb set 0 (n-tiles 0..3)     - sub-steps 0 and 2 of a K step
  v[72:87]    fa set 1                  v[88:103] fb set 1                    - sub-steps 1 and 3
  v[104:107]  the four A-fragment addresses of the sub-step being read
  v[0:39]     left to the compiler for the operands
Operands: qb{i} = byte address of patch pixel (tile row 4 wm + i, column lrow) in plane 0; sw{c}{k} = ((lgrp + 4k) ^ ((lrow + c) & 7)) << 4
(the swizzled slot of k half k for a pixel whose index is congruent lrow + c mod 8: tap (kh, kw) of m-tile i has c = (2i + 2kh + kw) & 7,
18 = 2 mod 8); rb{k} = byte address of this lane's B fragment row in buffer 0, plane 0, n-tile 0, k half k; vo{p} = DMA source offsets.

One K step t = tap (buffer c = t & 1):
  S0: 16 MFMAs (set 0) || reads of sub-step 1 -> set 1          S1: 16 MFMAs (set 1) || reads of sub-step 2 -> set 0
  S2: 16 MFMAs (set 0) || reads of sub-step 3 -> set 1          s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier   [tap t+1 landed; buffer c fully read]
  S3: 16 MFMAs (set 1) || 4 LDS-DMA requests of tap t+2 -> buffer c || reads of sub-step 0 of tap t+1 -> set 0
All nine steps are unrolled (the shifts, the swizzle classes and the offsets are immediates).
"""
import os
import sys

FA = [40, 72]
FB = [56, 88]
VT = 104
PLANE = 21 * 16 * 128          # 43008
RING = 2 * PLANE               # 86016: first weight buffer (32 KiB each: [plane 0: 128 rows x 128 B][plane 1])


ACC = 128


AGPR = True        # accumulators in a[0:63] (read back by bneck_agpr_tile<>) instead of v[128:191] + output operands


def mfma(op, i, j, s):
    if AGPR:
        a = 4 * (4 * i + j)
        return f"{op} a[{a}:{a+3}], v[{FA[s]+4*i}:{FA[s]+4*i+3}], v[{FB[s]+4*j}:{FB[s]+4*j+3}], a[{a}:{a+3}]"
    a = ACC + 4 * (4 * i + j)
    return f"{op} v[{a}:{a+3}], v[{FA[s]+4*i}:{FA[s]+4*i+3}], v[{FB[s]+4*j}:{FB[s]+4*j+3}], v[{a}:{a+3}]"


def reads(tap, ks, s):
    """the 8 fragment reads of sub-step ks of tap `tap` into register set s: (address add +) A read per m-tile, B read per n-tile"""
    kh, kw = divmod(tap, 3)
    d = 18 * kh + kw
    h, k = ks >> 1, ks & 1
    c = tap & 1
    out = []
    for i in range(4):
        cls = (2 * i + 2 * kh + kw) & 7
        out.append([f"v_add_u32 v{VT+i}, %[qb{i}], %[sw{cls}{k}]",
                    f"ds_read_b128 v[{FA[s]+4*i}:{FA[s]+4*i+3}], v{VT+i} offset:{d*128 + h*PLANE}"])
    for j in range(4):
        out.append([f"ds_read_b128 v[{FB[s]+4*j}:{FB[s]+4*j+3}], %[rb{k}] offset:{c*32768 + h*16384 + j*2048}"])
    # interleave A and B reads: A0 B0 A1 B1 ...
    return [out[0], out[4], out[1], out[5], out[2], out[6], out[3], out[7]]


def dmas(tap):
    """the four LDS-DMA requests of tap `tap`'s weights (128 rows x 256 B) into buffer tap & 1; %[sof] = tap * 256 already"""
    c = tap & 1
    out = []
    for p in range(4):
        out.append([f"s_add_u32 m0, %[wb], {c*32768 + (p >> 1)*16384 + (p & 1)*8192}", "s_nop 0",
                    f"buffer_load_dwordx4 %[vo{p}], %[srd], %[sof] offen lds"])
    return out


def interleave(mf, fillers):
    by = {}
    for slot, lines in fillers:
        by.setdefault(slot, []).extend(lines)
    out = list(by.get(-1, []))
    for s, m in enumerate(mf):
        out.append(m)
        out += by.get(s, [])
    for s in sorted(by):
        if s >= len(mf):
            out += by[s]
    return out


def substep(op, s, fillers):
    mf = [mfma(op, i, j, s) for i in range(4) for j in range(4)]
    return interleave(mf, fillers)


def step(op, tap, g):
    out = []
    for ks in range(3):                                   # S0, S1, S2: reads of the next sub-step in the first half of the slots
        f = [(k, r) for k, r in enumerate(reads(tap, ks + 1, (ks + 1) & 1))]           # slots 0..7
        out += substep(op, ks & 1, f)
        if ks < 2:
            out += ["s_waitcnt lgkmcnt(0)"]
    out += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    f = []
    if tap + 2 <= 8:
        d0 = 0 if g == 0 else 2
        f += [(-1, ["s_add_u32 %[sof], %[sof], 256"])]
        f += [(d0 + 4 * k, d) for k, d in enumerate(dmas(tap + 2))]
    if tap + 1 <= 8:
        f += [(1 + 2 * k, r) for k, r in enumerate(reads(tap + 1, 0, 0))][:8]
    out += substep(op, 1, f)
    out += ["s_waitcnt lgkmcnt(0)"]
    return out


def loop_text(op, g):
    L = ["s_mov_b32 %[m0s], m0"]
    # prologue: tap 0 is in buffer 0 (requested at kernel entry, waited for by the caller); request tap 1, zero the accumulators, read tap 0's sub-step 0
    L += ["s_add_u32 %[sof], %[sof], 256"]
    for d in dmas(1):
        L += d
    for r in reads(0, 0, 0):
        L += r
    for a in range(64):
        L.append(f"v_accvgpr_write_b32 a{a}, 0" if AGPR else f"v_mov_b32 v{ACC+a}, 0")
    L += ["s_waitcnt lgkmcnt(0)"]
    for tap in range(9):
        L += step(op, tap, g)
    L += ["s_nop 15", "s_nop 15"]
    if not AGPR:
        for t in range(32):
            L.append(f"v_pk_mov_b32 %[o{t}], v[{ACC+2*t}:{ACC+2*t+1}], v[{ACC+2*t}:{ACC+2*t+1}] op_sel:[0,1]")
    L += ["s_mov_b32 m0, %[m0s]"]
    return L


def emit(f):
    f.write("// GENERATED by tools/gen_bneck_asm.py - do not edit (register map, schedule: that file)\n")
    f.write("template <int TIDX> __device__ __forceinline__ f32x4_t bneck_agpr_tile();\n")
    for t in range(16):
        f.write(f"template <> __device__ __forceinline__ f32x4_t bneck_agpr_tile<{t}>() {{ float a, b, c, d; "
                f'asm volatile("v_accvgpr_read_b32 %0, a{4*t}\\n\\tv_accvgpr_read_b32 %1, a{4*t+1}\\n\\tv_accvgpr_read_b32 %2, a{4*t+2}\\n\\tv_accvgpr_read_b32 %3, a{4*t+3}" '
                f': "=v"(a), "=v"(b), "=v"(c), "=v"(d)); return f32x4_t{{a, b, c, d}}; }}\n')
    f.write("\n")
    for tname, op in (("BF16", "v_mfma_f32_16x16x32_bf16"), ("F16", "v_mfma_f32_16x16x32_f16")):
        for g in (0, 1):
            f.write(f"#define BNECK_ASM_LOOP_{tname}_G{g} \\\n")
            f.write(" \\\n".join(f'  "{l}\\n\\t"' for l in loop_text(op, g)))
            f.write("\n\n")


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multispectral-object-detection_amd", "csrc", "probes", "bottleneck_asm.inc")
    with open(out if len(sys.argv) < 2 else sys.argv[1], "w") as f:
        emit(f)
