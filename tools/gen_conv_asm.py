#!/usr/bin/env python3
"""Writes multispectral-object-detection_amd/csrc/conv_gemm_asm.inc: the hand-scheduled K loop of conv_gemm_asm_kernel
(csrc/conv_gemm_asm.hip) as inline-asm text, one string per (operand type, masked / unmasked staging, wave group).

Why a generator: hipcc re-orders whatever a HIP source says about a K step (profiles/r04_gemm_experiments.md: the
"register-double-buffered" 8-wave kernel came out as [5 reads, wait, 4 MFMAs] bursts with its eight LDS-DMA requests in one
block), so the whole main loop is ONE asm statement and every ds_read_b128 / buffer_load ... lds / s_waitcnt has a chosen
place between the 64 MFMAs of a K step.  Schedules were measured with tools/micro/kloop_asm (profiles/r06_kloop_microbench.md):
2 140 cycles per 256 x 256 x 64 step against the 2 048 the matrix pipe needs; 1.26-1.30 us at the clock this load holds
(the no-scheduling loop: 1.48).

Register map of the asm block (per lane; 8 waves of 128 x 64, two per SIMD: 256 registers each):
  a[0:127]    accumulators, acc[i][j] = a[4*(4i+j) : +3]   (i: 8 m-tiles of 16 rows, j: 4 n-tiles of 16 columns)
  v[32:63]    fa0[i]  A fragments of k half 0        v[64:79]    fb0[j]
  v[80:111]   fa1[i]  k half 1                       v[112:127]  fb1[j]
  v[27]       scratch, v[28:31] this step's four A voffsets after the tap mask (masked form)
  v[0:26]     left to the compiler for the operands
  s[92:95], s[96:99]   K-walk table entries {A byte offset, B byte offset, tap bit, -} of the step whose requests go out next, by step parity
LDS: [A image of even steps 32 KiB][A image of odd steps][B image even][B image odd], an image = 256 rows x 128 B, row r slot s =
k-granule s ^ (r & 7) (the image of conv_gemm_kernel: same fetches, same fragment reads, same k order per accumulator -> bit-identical
results).  The two buffers of an operand are 32 KiB apart so that one base register per (operand, k half) reaches both through the
16-bit offset field of ds_read_b128.

One K step t (buffer c = t & 1):
  half 0:  32 MFMAs on (fa0, fb0) || table entry t+3 -> quad[c^1] || 12 ds_read_b128 of k half 1 of buffer c -> (fa1, fb1)
           || (masked) the four A voffsets of step t+2: in-image for this tap ? offset : out of range (the DMA then writes zeros)
           s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier      [step t+1 has landed everywhere; every wave is done reading buffer c]
  half 1:  32 MFMAs on (fa1, fb1) || 8 LDS-DMA requests of step t+2 -> buffer c || 12 ds_read_b128 of k half 0 of buffer c^1
           s_waitcnt lgkmcnt(0)
The two wave groups (waves 0-3 / 4-7: a SIMD holds one wave of each) place their requests two MFMA slots apart.
"""
import os
import sys

TILES = [(8, 8), (7, 7), (7, 6), (6, 6), (4, 4)]      # (m-tiles of wave group 0, of group 1): tile heights 256, 224, 208, 192, 128
FA = [32, 80]
FB = [64, 112]
VO = 28          # v[28:31]: masked A voffsets
VT = 27          # scratch
QUAD = [92, 96]  # s[92:95] / s[96:99]
OOB = "%[voob]"   # a VGPR holding 0x80000000 (a literal beside vcc violates the constant-bus limit)


PAD_NOPS = int(os.environ.get("CONV_ASM_PAD_NOPS", "0"))     # experiment (profiles/r06_asm_kloop.md section 5): idle issue slots after every MFMA - same work, more cycles


def mfma(op, i, j, h):
    a = 4 * (4 * i + j)
    m = f"{op} a[{a}:{a+3}], v[{FA[h]+4*i}:{FA[h]+4*i+3}], v[{FB[h]+4*j}:{FB[h]+4*j+3}], a[{a}:{a+3}]"
    n = PAD_NOPS
    while n > 0:
        m += "\\n\\t" + f"s_nop {min(n, 16) - 1}"
        n -= 16
    return m


def reads(h, c, mt=8):
    out = []
    for j in range(4):
        out.append(f"ds_read_b128 v[{FB[h]+4*j}:{FB[h]+4*j+3}], %[rb{h}] offset:{c*32768 + j*2048}")
    for i in range(mt):
        out.append(f"ds_read_b128 v[{FA[h]+4*i}:{FA[h]+4*i+3}], %[ra{h}] offset:{c*32768 + i*2048}")
    return out


def mask_ops(q, masked, ap=4):
    """v[28+i] = (amask_i & tapbit) ? voa_i : OOB, tapbit = s[q+2]; one list per A row pass"""
    if not masked:
        return []
    return [[f"v_and_b32 v{VT}, s{q+2}, %[am{i}]", f"v_cmp_ne_u32 vcc, 0, v{VT}", f"v_cndmask_b32 v{VO+i}, {OOB}, %[voa{i}], vcc"] for i in range(ap)]


def dmas(c, q, masked, ap=4):
    """the ap + 4 LDS-DMA requests of one K step into buffer c (ap 64-row passes of the A image, 4 of the B image), offsets from table quad s[q:q+3]"""
    out = []
    for i in range(ap):
        vo = f"v{VO+i}" if masked else f"%[voa{i}]"
        out.append([f"s_add_u32 m0, %[wb], {c*32768 + i*8192}", "s_nop 0", f"buffer_load_dwordx4 {vo}, %[srda], s{q} offen lds"])
    for i in range(4):
        out.append([f"s_add_u32 m0, %[wb], {65536 + c*32768 + i*8192}", "s_nop 0", f"buffer_load_dwordx4 %[vob{i}], %[srdb], s{q+1} offen lds"])
    return out


def table_load(q):
    return [f"s_load_dwordx4 s[{q}:{q+3}], %[tab], %[toff]", "s_add_u32 %[toff], %[toff], 16"]


def interleave(mf, fillers):
    """fillers: list of (slot, lines); lines go right AFTER MFMA slot (slot -1: before the first MFMA)"""
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


def step(op, c, g, masked, mt=8, ap=4):
    order = [(i, j) for i in range(mt) for j in range(4)]
    mf0 = [mfma(op, i, j, 0) for (i, j) in order]
    mf1 = [mfma(op, i, j, 1) for (i, j) in order]
    n, nr, nd = 4 * mt, mt + 4, ap + 4
    q = QUAD[c]
    f0 = [(-1, table_load(QUAD[c ^ 1]))]
    f0 += [(k, [r]) for k, r in enumerate(reads(1, c, mt))]                   # slots 0 .. mt+3
    ms = max(1, (n - nr - 2) // max(ap, 1))
    f0 += [(nr + 2 + ms * k, m) for k, m in enumerate(mask_ops(q, masked, ap))]   # behind the reads, spread over the rest of the half
    out = interleave(mf0, f0)
    out += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    ds = 4 if n >= 4 * nd else (3 if n >= 3 * nd else 2)                      # one request per ds MFMA slots; the two wave groups ds/2 slots apart
    d0 = 0 if g == 0 else ds // 2
    f1 = [(d0 + ds * k, d) for k, d in enumerate(dmas(c, q, masked, ap))]
    f1 += [(1 + 2 * k, [r]) for k, r in enumerate(reads(0, c ^ 1, mt))]       # odd slots
    out += interleave(mf1, f1)
    out += ["s_waitcnt lgkmcnt(0)"]
    return out


def loop_text(op, g, masked, mt=8, ap=4):
    L = ["s_mov_b32 %[m0s], m0"]
    # prologue: table entries 0 / 1 -> requests of steps 0 / 1; entry 2 -> quad 0; (accumulators zeroed under the requests' flight;) wait
    # for step 0; read its k half 0
    L += table_load(QUAD[0]) + table_load(QUAD[1]) + ["s_waitcnt lgkmcnt(0)"]
    for c in (0, 1):
        for m in mask_ops(QUAD[c], masked, ap):
            L += m
        for d in dmas(c, QUAD[c], masked, ap):
            L += d
    L += table_load(QUAD[0])
    for a in range(16 * mt):
        L.append(f"v_accvgpr_write_b32 a{a}, 0")
    L += [f"s_waitcnt vmcnt({ap + 4})", "s_barrier"]
    L += reads(0, 0, mt)
    L += ["s_waitcnt lgkmcnt(0)"]
    # the loop computes its masked offsets in half 0 of each step from quad[c]: entry 2 is in quad 0 now
    L += [".p2align 6", "1:"]
    L += step(op, 0, g, masked, mt, ap)
    L += step(op, 1, g, masked, mt, ap)
    L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b"]
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_mov_b32 m0, %[m0s]"]
    return L


def chain2_text(op):
    """One K step (64 channels = one LDS image) of the chained kernel's SECOND GEMM: A fragments from the image at %[ca0] / %[ca1] (k half 0 / 1),
    B fragments from the weight buffer at %[cb0] / %[cb1]; no staging inside (the caller streams the weights between the steps)."""
    order = [(i, j) for i in range(8) for j in range(4)]
    L = []
    for j in range(4):
        L.append(f"ds_read_b128 v[{FB[0]+4*j}:{FB[0]+4*j+3}], %[cb0] offset:{j*2048}")
    for i in range(8):
        L.append(f"ds_read_b128 v[{FA[0]+4*i}:{FA[0]+4*i+3}], %[ca0] offset:{i*2048}")
    L.append("s_waitcnt lgkmcnt(0)")
    r1 = [f"ds_read_b128 v[{FB[1]+4*j}:{FB[1]+4*j+3}], %[cb1] offset:{j*2048}" for j in range(4)] + \
         [f"ds_read_b128 v[{FA[1]+4*i}:{FA[1]+4*i+3}], %[ca1] offset:{i*2048}" for i in range(8)]
    L += interleave([mfma(op, i, j, 0) for (i, j) in order], [(k, [r]) for k, r in enumerate(r1)])
    L.append("s_waitcnt lgkmcnt(0)")
    L += [mfma(op, i, j, 1) for (i, j) in order]
    L += ["s_nop 7"]
    return L


def emit(f):
    f.write("// GENERATED by tools/gen_conv_asm.py - do not edit (register map, schedule and the reasons: that file)\n")
    # accumulator tile T of this lane (a[4T : 4T+3]) out of the accumulator file, one specialisation per tile (register names are text)
    f.write("template <int TIDX> __device__ __forceinline__ f32x4_t agpr_tile();\n")
    for t in range(32):
        f.write(f"template <> __device__ __forceinline__ f32x4_t agpr_tile<{t}>() {{ float a, b, c, d; "
                f'asm volatile("v_accvgpr_read_b32 %0, a{4*t}\\n\\tv_accvgpr_read_b32 %1, a{4*t+1}\\n\\tv_accvgpr_read_b32 %2, a{4*t+2}\\n\\tv_accvgpr_read_b32 %3, a{4*t+3}" '
                f': "=v"(a), "=v"(b), "=v"(c), "=v"(d)); return f32x4_t{{a, b, c, d}}; }}\n')
    f.write("\n")
    f.write("#define CONV_ASM_ZERO_ACC \\\n" + " \\\n".join(f'  "v_accvgpr_write_b32 a{a}, 0\\n\\t"' for a in range(128)) + "\n\n")
    for tname, op in (("BF16", "v_mfma_f32_16x16x32_bf16"), ("F16", "v_mfma_f32_16x16x32_f16")):
        f.write(f"#define CONV_ASM_CHAIN2_{tname} \\\n")
        f.write(" \\\n".join(f'  "{l}\\n\\t"' for l in chain2_text(op)))
        f.write("\n\n")
    # one text per (operand type, masked / unmasked staging, m-tiles of the wave group, A passes of the tile): the tile is 16 (MT0 + MT1) rows,
    # wave group 0 (waves 0-3) owns the first MT0 m-tiles, group 1 the next MT1 (a SIMD holds one wave of each: MT0 + MT1 MFMA rows per SIMD)
    done = set()
    for tname, op in (("BF16", "v_mfma_f32_16x16x32_bf16"), ("F16", "v_mfma_f32_16x16x32_f16")):
        for mname, masked in (("MASK", True), ("NOMASK", False)):
            for (mt0, mt1) in TILES:
                ap = (16 * (mt0 + mt1) + 63) // 64
                for g, mt in ((0, mt0), (1, mt1)):
                    name = f"CONV_ASM_LOOP_{tname}_{mname}_M{mt}_P{ap}_G{g}"
                    if name in done:
                        continue
                    done.add(name)
                    f.write(f"#define {name} \\\n")
                    f.write(" \\\n".join(f'  "{l}\\n\\t"' for l in loop_text(op, g, masked, mt, ap)))
                    f.write("\n\n")


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multispectral-object-detection_amd", "csrc", "conv_gemm_asm.inc")
    with open(out if len(sys.argv) < 2 else sys.argv[1], "w") as f:
        emit(f)
