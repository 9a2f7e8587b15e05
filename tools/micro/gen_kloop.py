#!/usr/bin/env python3
"""Generator of the hand-scheduled K loop of the 8-wave 256 x 256 x 64 GEMM tile (round 6).

hipcc re-orders whatever a HIP source says about the K step (profiles/r04_gemm_experiments.md: the "register-double-buffered"
8-wave kernel came out as [5 reads, wait, 4 MFMAs] bursts with the eight LDS-DMA requests in one block), so the whole main loop
is ONE `asm volatile` statement whose text this script writes: every ds_read_b128, every buffer_load ... lds and every
s_waitcnt sits at a chosen position between the 64 MFMAs of a K step, per wave GROUP (waves 0-3 / 4-7 share the SIMDs pairwise,
so the two groups get different placements and their LDS-DMA issue stalls do not coincide).

Register map of the asm block (per lane):
  a[0:127]    accumulators, acc[i][j] = a[4*(4i+j) : +3]   (i: 8 m-tiles of the 128-row wave tile, j: 4 n-tiles of its 64 columns)
  v[32:63]    fa0[i]   fragments of the A operand, k half 0      v[64:79]    fb0[j]
  v[80:111]   fa1[i]   k half 1                                  v[112:127]  fb1[j]
  v[0:31]     left to the compiler for the operands (8 DMA voffsets, 8 fragment-read base addresses)
LDS: two 64-KiB K-step buffers, each [A image 256 rows x 128 B][B image 256 rows x 128 B], row r slot s = k-granule s ^ (r & 7).

One K step t (buffer c = t & 1), the schedule every variant shares:
  half 0:  32 MFMAs on (fa0, fb0)   ||  12 ds_read_b128 of k half 1 of buffer c -> (fa1, fb1)
           s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier     [step t+1 landed everywhere; every wave is done reading buffer c]
  half 1:  32 MFMAs on (fa1, fb1)   ||  8 LDS-DMA requests of step t+2 -> buffer c  ||  12 ds_read_b128 of k half 0 of buffer c^1 -> (fa0, fb0)
           s_waitcnt lgkmcnt(0)
Usage: gen_kloop.py > kloop_variants.inc
"""
import sys

FA = [32, 80]     # first VGPR of fa0 / fa1
FB = [64, 112]    # first VGPR of fb0 / fb1


def mfma(i, j, h, ablate):
    if ablate & 2:
        return []
    a = 4 * (4 * i + j)
    return [f"v_mfma_f32_16x16x32_bf16 a[{a}:{a+3}], v[{FA[h]+4*i}:{FA[h]+4*i+3}], v[{FB[h]+4*j}:{FB[h]+4*j+3}], a[{a}:{a+3}]"]


def reads(h, c, ablate):
    """the 12 fragment reads of k half h from buffer c, B fragments first (the first MFMAs need all of them)"""
    if ablate & 4:
        return []
    out = []
    for j in range(4):
        out.append(f"ds_read_b128 v[{FB[h]+4*j}:{FB[h]+4*j+3}], %[rb{h}{c}] offset:{j*2048}")
    for i in range(8):
        out.append(f"ds_read_b128 v[{FA[h]+4*i}:{FA[h]+4*i+3}], %[ra{h}{c}] offset:{i*2048}")
    return out


def dmas(c, ablate, wrap=True):
    """the 8 LDS-DMA requests of one K step into buffer c (each: M0 = LDS address of this wave's 1-KiB piece, then the request),
    followed by the scalar advance of the two K offsets"""
    if ablate & 1:
        return []
    out = []
    for i in range(4):
        out.append([f"s_add_u32 m0, %[wb], {c*65536 + i*8192}", "s_nop 0",
                    f"buffer_load_dwordx4 %[voa{i}], %[srda], %[sofa] offen lds"])
    for i in range(4):
        out.append([f"s_add_u32 m0, %[wb], {c*65536 + 32768 + i*8192}", "s_nop 0",
                    f"buffer_load_dwordx4 %[vob{i}], %[srdb], %[sofb] offen lds"])
    adv = ["s_add_u32 %[sofa], %[sofa], 128", "s_add_u32 %[sofb], %[sofb], 128"]
    if wrap:
        adv += ["s_cmp_ge_u32 %[sofa], %[kwrap]", "s_cselect_b32 %[sofa], 0, %[sofa]", "s_cselect_b32 %[sofb], 0, %[sofb]"]
    out[-1] = out[-1] + adv
    return out


def interleave(mf, fillers, pos):
    """mf: list of MFMA lines (one per slot); fillers[k] (a line or a list of lines) goes right AFTER MFMA slot pos[k]
    (pos -1: before the first MFMA).  Fillers beyond the last slot are appended."""
    out = []
    by = {}
    for k, f in enumerate(fillers):
        by.setdefault(pos[k], []).append(f)
    for f in by.get(-1, []):
        out += f if isinstance(f, list) else [f]
    n = len(mf) if mf else 32
    for s in range(n):
        if mf:
            out.append(mf[s])
        for f in by.get(s, []):
            out += f if isinstance(f, list) else [f]
    for p in sorted(by):
        if p >= n:
            for f in by[p]:
                out += f if isinstance(f, list) else [f]
    return out


def step(c, sched, ablate):
    """one K step on buffer c"""
    if sched.get("jouter"):
        order = [(i, j) for j in range(4) for i in range(8)]
    else:
        order = [(i, j) for i in range(8) for j in range(4)]
    mf0 = [l for (i, j) in order for l in mfma(i, j, 0, ablate)]
    mf1 = [l for (i, j) in order for l in mfma(i, j, 1, ablate)]
    r1 = reads(1, c, ablate)
    r0 = reads(0, c ^ 1, ablate)
    d = dmas(c, ablate)
    out = []
    if sched.get("prio"):
        out.append("s_setprio 1")
    out += interleave(mf0, r1, [sched["r1_start"] + k * sched["r1_stride"] for k in range(len(r1))])
    out += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    fill = d + r0
    pos = [sched["d_start"] + k * sched["d_stride"] for k in range(len(d))] + \
          [sched["r0_start"] + k * sched["r0_stride"] for k in range(len(r0))]
    out += interleave(mf1, fill, pos)
    out += ["s_waitcnt lgkmcnt(0)"]
    return out


def loop_text(sched, ablate):
    L = []
    L.append("s_mov_b32 %[m0s], m0")
    for a in range(128):
        L.append(f"v_accvgpr_write_b32 a{a}, 0")
    # prologue: request steps 0 and 1, wait for step 0, read its k half 0
    for c in (0, 1):
        for f in dmas(c, 0):
            L += f
    L += ["s_waitcnt vmcnt(8)", "s_barrier"]
    L += reads(0, 0, 0)
    L += ["s_waitcnt lgkmcnt(0)", ".p2align 6", "1:"]
    L += step(0, sched, ablate)
    L += step(1, sched, ablate)
    L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b"]
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_mov_b32 m0, %[m0s]"]
    return L


# schedule variants: (name, {group 0 placement}, {group 1 placement}, ablate)
def S(r1_start=0, r1_stride=1, d_start=0, d_stride=2, r0_start=16, r0_stride=1, prio=0, jouter=0):
    return dict(r1_start=r1_start, r1_stride=r1_stride, d_start=d_start, d_stride=d_stride, r0_start=r0_start, r0_stride=r0_stride, prio=prio, jouter=jouter)


D4A = dict(d_start=0, d_stride=4, r0_start=1, r0_stride=2)
D4B = dict(d_start=2, d_stride=4, r0_start=1, r0_stride=2)
VARIANTS = [
    # name, group-0 schedule, group-1 schedule, ablate (1 = no DMA, 2 = no MFMA, 4 = no fragment reads)
    ("same_d2", S(), S(), 0),                                                       # both groups: DMAs after MFMAs 0,2,..14, reads from 16
    ("stag_d4", S(**D4A), S(**D4B), 0),                                             # DMAs every 4th slot, the groups 2 slots apart, reads on the odd slots
    ("stag_d4_r1s2", S(r1_stride=2, **D4A), S(r1_start=1, r1_stride=2, **D4B), 0),  # + half-0 reads every second slot
    ("stag_d4_prio1", S(**D4A), S(prio=1, **D4B), 0),                               # + static priority for the second-dispatched group
    ("stag_d4_prio0", S(prio=1, **D4A), S(**D4B), 0),                               # + static priority for the first group
    ("stag_d4_jout", S(jouter=1, **D4A), S(jouter=1, **D4B), 0),                    # + MFMA order j outer (B fragment reused by 8 consecutive MFMAs)
    ("stag_d3", S(d_start=0, d_stride=3, r0_start=1, r0_stride=3), S(d_start=2, d_stride=3, r0_start=1, r0_stride=3), 0),
    ("stag_d4_r0late", S(d_start=0, d_stride=4, r0_start=2, r0_stride=2), S(d_start=2, d_stride=4, r0_start=4, r0_stride=2), 0),
    ("half_split", S(d_start=0, d_stride=2, r0_start=16, r0_stride=1), S(d_start=16, d_stride=2, r0_start=0, r0_stride=1), 0),   # group 0 DMAs in the first 16 slots, group 1 in the last 16
    ("d4_same", S(**D4A), S(**D4A), 0),                                             # the d4 placement without the stagger
    ("mfma_only", S(), S(), 1 | 4),
    ("mfma_dma_d4", S(**D4A), S(**D4B), 4),
]


def emit():
    print("// GENERATED by tools/micro/gen_kloop.py - do not edit; see that file for the register map and the schedule")
    print(f"#define KLOOP_NVARIANTS {len(VARIANTS)}")
    print("static const char* kloop_names[KLOOP_NVARIANTS] = {" + ", ".join(f'"{v[0]}"' for v in VARIANTS) + "};")
    print("#define KLOOP_FOR_EACH(X) " + " ".join(f"X({n})" for n in range(len(VARIANTS))))
    for n, (name, s0, s1, abl) in enumerate(VARIANTS):
        for grp, s in ((0, s0), (1, s1)):
            print(f"#define KLOOP_ASM_{n}_G{grp} \\")
            lines = loop_text(s, abl)
            print(" \\\n".join(f'  "{l}\\n\\t"' for l in lines))
            print()


if __name__ == "__main__":
    emit()
