"""Generator of scripts/ubench/pipe_ubench.hip: two hardware questions behind DESIGN.md section 9, answered with
register-exact inline-asm loops (hipcc cannot allocate 320 accumulators: it parks them all in AGPRs and spills).

 1. Tile economy of the conv / GEMM kernel (power-limited at 57-61 % matrix-pipe duty, 1.5-1.6 GHz): what do the LDS fragment
    reads cost?  Per k16 step a wave with an MT x NT tile of 32 x 32 blocks issues MT * NT MFMAs and MT + NT ds_read_b128:
      8 waves, 2 x 5  (today's 256 x 320 tile: 7 reads per 10 MFMAs)
      4 waves, 4 x 5  (one wave per SIMD, 128 x 160 wave tile, 320 accumulators: 9 per 20)
      4 waves, 4 x 4  (256 accumulators: 8 per 16)
    each with and without the reads (fragments then stay in registers), and with the LDS-DMA refill stream of the real kernel
    (18 KiB per k16 step and workgroup, from an L2-resident window). Printed: chip TFLOP/s, shader clock, matrix-pipe duty.
 2. The attention kernel's per-tile mix (14 MFMAs, 32 v_exp_f32, 60 plain VALU per wave and 64-key tile): SQ counters say a
    SIMD spends VALU time + MFMA time per wave-tile. Do the two pipes overlap (a) inside one wave, (b) across the two waves
    of a SIMD?  Kernels: MFMA only, VALU only, both in every wave, roles split between the waves of a SIMD.

   python scripts/ubench/gen_pipe_ubench.py > scripts/ubench/pipe_ubench.hip
   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o scripts/ubench/pipe_ubench scripts/ubench/pipe_ubench.hip
"""
import sys

out = []
emit = out.append

FRAG0 = 8          # first fragment VGPR
ACCV0 = 128        # accumulator tiles beyond the 16 that fit the AGPRs live in v[128..]


def acc_reg(n, agpr):
    """4-wave kernels (512 registers per wave): tiles 0-15 in the AGPRs, the rest in v[128..]; 8-wave kernels (256 per wave, one
    unified budget): every tile in VGPRs from v64, as hipcc allocates the product kernel (248 VGPRs, no AGPRs)."""
    if not agpr:
        return "v[%d:%d]" % (64 + 16 * n, 64 + 16 * n + 15)
    return "a[%d:%d]" % (16 * n, 16 * n + 15) if n < 16 else "v[%d:%d]" % (ACCV0 + 16 * (n - 16), ACCV0 + 16 * (n - 16) + 15)


def frag(s, idx, per_set):
    r = FRAG0 + (s * per_set + idx) * 4
    return "v[%d:%d]" % (r, r + 3)


def tile_kernel(name, waves, mt, nt, reads, fill):
    """One k16 step = mt*nt MFMAs; fragment set s^1 is read (mt+nt ds_read_b128) while set s feeds the MFMAs."""
    per_set = mt + nt
    nacc = mt * nt
    lines = []
    a = lines.append
    # prelude: zero accumulators, load both fragment sets
    agpr = waves == 4
    for n in range(nacc):
        for r in range(16):
            if agpr and n < 16:
                a("v_accvgpr_write_b32 a%d, 0" % (16 * n + r))
            elif agpr:
                a("v_mov_b32 v%d, 0" % (ACCV0 + 16 * (n - 16) + r))
            else:
                a("v_mov_b32 v%d, 0" % (64 + 16 * n + r))
    for s in range(2):
        for i in range(per_set):
            a("ds_read_b128 %s, %%[addr] offset:%d" % (frag(s, i, per_set), (s * per_set + i) * 1024))
    a("s_waitcnt lgkmcnt(0)")
    if fill:
        a("s_mov_b32 m0, %[m0v]")
    a("s_mov_b32 s20, %[iters]")
    a("L_loop_%=:")
    fill_per_wave = (18 // waves) if fill else 0   # 1-KiB LDS-DMA instructions per wave and k16 step (18 KiB per workgroup)
    if fill and fill_per_wave == 0:
        fill_per_wave = 1
    for s in range(2):
        # MFMAs of set s; the reads of the step after next go to set s as soon as its MFMAs have issued - instead, as
        # the product kernel does, read the OTHER set's replacement now: set s^1 was consumed in the previous half
        rd = 0
        fl = 0
        n = 0
        for i in range(mt):
            for j in range(nt):
                a("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (acc_reg(n, agpr), frag(s, i, per_set), frag(s, mt + j, per_set), acc_reg(n, agpr)))
                n += 1
                # spread reads and refill instructions between the MFMAs
                if reads and rd < per_set and (n * per_set) // nacc > rd:
                    a("ds_read_b128 %s, %%[addr] offset:%d" % (frag(s ^ 1, rd, per_set), ((s ^ 1) * per_set + rd) * 1024))
                    rd += 1
                if fill and fl < fill_per_wave and (n * fill_per_wave) // nacc > fl:
                    a("buffer_load_dwordx4 %%[voff], %%[rsrc], 0 offen offset:%d lds" % (fl * 1024 % 4096))
                    fl += 1
        while reads and rd < per_set:
            a("ds_read_b128 %s, %%[addr] offset:%d" % (frag(s ^ 1, rd, per_set), ((s ^ 1) * per_set + rd) * 1024))
            rd += 1
        if reads:
            a("s_waitcnt lgkmcnt(0)")
        if fill:
            a("s_waitcnt vmcnt(%d)" % fill_per_wave)
    a("s_sub_u32 s20, s20, 1")
    a("s_cmp_lg_u32 s20, 0")
    a("s_cbranch_scc1 L_loop_%=")
    if fill:
        a("s_waitcnt vmcnt(0)")
    a("s_nop 15")
    a("s_nop 15")
    a("v_accvgpr_read_b32 %[res], a0" if agpr else "v_mov_b32 %[res], v64")
    nfrag = 2 * per_set * 4
    clob = ["v%d" % r for r in range(FRAG0, FRAG0 + nfrag)]
    if agpr:
        clob += ["a%d" % r for r in range(min(nacc, 16) * 16)]
        clob += ["v%d" % r for r in range(ACCV0, ACCV0 + max(0, nacc - 16) * 16)]
    else:
        assert FRAG0 + nfrag <= 64 and 64 + 16 * nacc <= 248
        clob += ["v%d" % r for r in range(64, 64 + 16 * nacc)]
    clob += ["s20", "scc", "memory"]   # m0 is written too: reserved, hipcc reloads it before its own uses
    assert FRAG0 + nfrag <= ACCV0
    body = "\n".join('      "%s\\n"' % l for l in lines)
    emit("""
__global__ __launch_bounds__(%d) void %s(int iters, const char* src, float* sink, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  init_lds(smem, 24576);
  __syncthreads();
  const unsigned addr = (unsigned)(lane * 16);
  const unsigned voff = (unsigned)(((blockIdx.x * %d + wave) * 4096 + lane * 16) & 0x1fffff);
  const u32x4 rsrc = make_rsrc(src);
  const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane(24576 + wave * 4096);
  float res;
  const unsigned long long t0 = memtime(), r0 = memrealtime();
  asm volatile(
%s
      : [res] "=v"(res)
      : [addr] "v"(addr), [iters] "s"(iters), [voff] "v"(voff), [rsrc] "s"(rsrc), [m0v] "s"(m0v)
      : %s);
  const unsigned long long t1 = memtime(), r1 = memrealtime();
  if (lane == 0) { atomicMax(&clk[2 * blockIdx.x], t1 - t0); atomicMax(&clk[2 * blockIdx.x + 1], r1 - r0); }
  if (res == 123.456f) sink[0] = res;
}
""" % (64 * waves, name, waves, body, ", ".join('"%s"' % c for c in clob)))
    return dict(name=name, waves=waves, mfma_per_iter=2 * nacc, reads=2 * per_set if reads else 0,
                fill=2 * fill_per_wave * waves if fill else 0, lds=100 * 1024)


def kloop_kernel(name, waves, mt, nt, rd, sync, swz, shared_b=False):
    """Ablation ladder from the ideal loop towards conv_gemm.hip's K loop (BK = 64, 2-deep ring). One loop iteration = one K
    step of 64 = 4 k16 segments; fragment sets alternate.
      rd   'spread': the next segment's fragment reads sit between this segment's MFMAs, one lgkmcnt(0) per segment
           'late'  : as hipcc schedules the product loop: the reads follow MFMA #(n-3) of the segment (their registers are
                     free only then) and the next segment's MFMAs wait for them one by one (lgkmcnt(k) in front of each)
      sync 'none'  : refill instructions spread over the K step, counted wait only
           'wait'  : at the product's sync point (middle of segment 3): s_waitcnt vmcnt(0), then the whole refill back to back
           'barrier': the same + s_barrier (the product)
      shared_b  the B part of the refill (5 of 9 pieces per wave: the weight tile) reads the SAME addresses in every workgroup, as
           the product does when one weight tile serves all row tiles (N = 320: every CU streams the whole weight matrix) -
           the A part stays private; otherwise every workgroup streams its own window
      swz  fragment reads use the product's LDS layout (128-byte rows, XOR-swizzled 16-byte chunks) instead of lane-linear 1-KiB blocks"""
    per_set, nacc = mt + nt, mt * nt
    agpr = waves == 4
    lpt = (256 // 8 + 320 // 8) // waves     # 1-KiB LDS-DMA instructions per wave and K step (256 x 320 tile: 72 KiB)
    lines = []
    a = lines.append
    for n in range(nacc):
        for r in range(16):
            if agpr and n < 16:
                a("v_accvgpr_write_b32 a%d, 0" % (16 * n + r))
            elif agpr:
                a("v_mov_b32 v%d, 0" % (ACCV0 + 16 * (n - 16) + r))
            else:
                a("v_mov_b32 v%d, 0" % (64 + 16 * n + r))

    def read(s, idx, ks):
        # idx: 0..mt-1 = A blocks, mt.. = B blocks
        if swz:
            return "ds_read_b128 %s, %%[ad%d] offset:%d" % (frag(s, idx, per_set), ks, idx * 4096)
        return "ds_read_b128 %s, %%[ad0] offset:%d" % (frag(s, idx, per_set), (ks * per_set + idx) * 1024 % 24576)

    # read order of a segment: A0, B0 .. B(nt-1), A1 ..: the first MFMAs need the fewest reads
    order = [0] + [mt + j for j in range(nt)] + list(range(1, mt))
    pos = {idx: k for k, idx in enumerate(order)}
    for idx in order:
        a(read(0, idx, 0))
    a("s_waitcnt lgkmcnt(0)")
    a("s_mov_b32 m0, %[m0v]")
    a("s_mov_b32 s20, %[iters]")
    a("L_loop_%=:")
    fills_done = 0
    for seg in range(4):
        s = seg & 1
        nks = (seg + 1) & 3
        rdi = 0
        have = -1   # highest read position of this segment's operands already waited for
        for n in range(nacc):
            i, j = divmod(n, nt)
            if rd == "late" and rdi == 0:   # (once this segment's own reads are in flight the count covers them too)
                need = max(pos[i], pos[mt + j])
                if need > have:
                    a("s_waitcnt lgkmcnt(%d)" % (per_set - 1 - need))
                    have = need
            a("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (acc_reg(n, agpr), frag(s, i, per_set), frag(s, mt + j, per_set), acc_reg(n, agpr)))
            if seg == 3 and n + 1 == nacc // 2:   # the product's sync point
                if sync == "none":
                    a("s_waitcnt vmcnt(%d)" % lpt)
                else:
                    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
                    if sync == "barrier":
                        a("s_barrier")
                    for f in range(lpt):
                        vo = "%[voffb]" if (shared_b and f >= lpt * 4 // 9) else "%[voff]"
                        a("buffer_load_dwordx4 %s, %%[rsrc], 0 offen offset:%d lds" % (vo, f * 1024 % 4096))
            if sync == "none":
                tot = (seg * nacc + n + 1) * lpt // (4 * nacc)
                while fills_done < tot:
                    a("buffer_load_dwordx4 %%[voff], %%[rsrc], 0 offen offset:%d lds" % (fills_done * 1024 % 4096))
                    fills_done += 1
            if rd == "spread":
                while rdi < per_set and ((n + 1) * per_set) // nacc > rdi:
                    a(read(s ^ 1, order[rdi], nks))
                    rdi += 1
            elif n + 1 >= nacc - 2:
                # after MFMA n-3 .. n-1: everything that is left, in three portions
                share = [per_set - 2 * (per_set // 3), per_set // 3, per_set // 3][n + 1 - (nacc - 2)] if n + 1 < nacc + 1 else 0
                for _ in range(share):
                    if rdi < per_set:
                        a(read(s ^ 1, order[rdi], nks))
                        rdi += 1
        while rdi < per_set:
            a(read(s ^ 1, order[rdi], nks))
            rdi += 1
        if rd == "spread":
            a("s_waitcnt lgkmcnt(0)")
    a("s_sub_u32 s20, s20, 1")
    a("s_cmp_lg_u32 s20, 0")
    a("s_cbranch_scc1 L_loop_%=")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a("s_nop 15")
    a("s_nop 15")
    a("v_accvgpr_read_b32 %[res], a0" if agpr else "v_mov_b32 %[res], v64")
    nfrag = 2 * per_set * 4
    clob = ["v%d" % r for r in range(FRAG0, FRAG0 + nfrag)]
    if agpr:
        clob += ["a%d" % r for r in range(min(nacc, 16) * 16)]
        clob += ["v%d" % r for r in range(ACCV0, ACCV0 + max(0, nacc - 16) * 16)]
    else:
        assert FRAG0 + nfrag <= 64 and 64 + 16 * nacc <= 248
        clob += ["v%d" % r for r in range(64, 64 + 16 * nacc)]
    clob += ["s20", "scc", "memory"]
    body = "\n".join('      "%s\\n"' % l for l in lines)
    emit("""
__global__ __launch_bounds__(%d) void %s(int iters, const char* src, float* sink, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  init_lds(smem, 40960);
  __syncthreads();
  // product layout: fragment row = lane & 31 (128-byte rows), chunk (2 ks + half) ^ ((row >> 1) & 7); lane-linear otherwise
  const int row = lane & 31, half = lane >> 5;
  const unsigned sw0 = (unsigned)(row * 128 + ((half ^ ((row >> 1) & 7)) * 16));
  const unsigned ad0 = %s, ad1 = ad0 ^ 32u, ad2 = ad0 ^ 64u, ad3 = ad0 ^ 96u;
  const unsigned voff = (unsigned)(((blockIdx.x * %d + wave) * 4096 + lane * 16) & 0x1fffff);
  const unsigned voffb = (unsigned)(0x200000 + wave * 4096 + lane * 16);  // the same 32 KiB for every workgroup
  const u32x4 rsrc = make_rsrc(src);
  const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane(40960 + wave * 4096);
  float res;
  const unsigned long long t0 = memtime(), r0 = memrealtime();
  asm volatile(
%s
      : [res] "=v"(res)
      : [ad0] "v"(ad0), [ad1] "v"(ad1), [ad2] "v"(ad2), [ad3] "v"(ad3), [iters] "s"(iters), [voff] "v"(voff), [voffb] "v"(voffb),
        [rsrc] "s"(rsrc), [m0v] "s"(m0v)
      : %s);
  const unsigned long long t1 = memtime(), r1 = memrealtime();
  if (lane == 0) { atomicMax(&clk[2 * blockIdx.x], t1 - t0); atomicMax(&clk[2 * blockIdx.x + 1], r1 - r0); }
  if (res == 123.456f) sink[0] = res;
}
""" % (64 * waves, name, "sw0" if swz else "(unsigned)(lane * 16)", waves, body, ", ".join('"%s"' % c for c in clob)))
    return dict(name=name, waves=waves, mfma_per_iter=4 * nacc, reads=4 * per_set, fill=lpt * waves, lds=100 * 1024)


def attn_kernel(name, barrier=True, order="product", staging=True, wps=4, data="zero", skip=()):
    """Skeleton of k_attention<48, 64, 40, 2, false, 32, 8> (attn.hip): per wave and 64-key tile two 32-key groups of
    3 K-fragment reads + 3 chained QK^T MFMAs -> maximum (v_med3 + 7 v_max3) -> 16 v_exp_f32 -> 8 packed converts ->
    4 V^T-fragment reads + 4 PV MFMAs, the next tile's K / V^T fetched to registers and committed to the other LDS buffer,
    one barrier per tile; 8-wave workgroups, `wps` waves per SIMD. The data are meaningless, the dependences are the kernel's.
      data  'zero': q = 0.5, scores - 50: every probability is 0 (the PV operands do not toggle); 'random': q, k, v random, scores
            in (-6, -2): probabilities in (0.015, 0.25) - the operand activity of the real kernel (power!)
      skip  subset of {'exp', 'pv', 'qk'}: instruction classes left out (timing / power experiments)
      order 'product': QK(g) softmax(g) PV(g) per group;  'qk_ahead': QK of both groups first (second score block: +16 VGPRs)
            'pv_behind': QK(g+1) is issued before softmax(g) and PV(g) after softmax(g+1) never crosses a tile"""
    S0, S1 = 36, 128
    lines = []
    a = lines.append
    if data == "random":
        for ks in range(3):
            a("ds_read_b128 v[%d:%d], %%[vad] offset:%d" % (8 + 4 * ks, 11 + 4 * ks, 14336 + ks * 32))
        a("s_waitcnt lgkmcnt(0)")
        a("s_mov_b32 s22, 0x30003000")   # q in +-[1/16, 1/8): 48-term dot products of a few tenths
        for r in range(8, 20):
            a("v_pk_mul_f16 v%d, v%d, s22" % (r, r))
        for r in range(20, 36):
            a("v_mov_b32 v%d, 0xc0800000" % r)   # -m = -4
    else:
        for r in range(8, 36):
            a("v_mov_b32 v%d, 0x38003800" % r) if r < 20 else a("v_mov_b32 v%d, 0xc2480000" % r)   # q = 0.5, -m = -50
    for r in range(88, 120):
        a("v_mov_b32 v%d, 0" % r)
    if order == "q64":
        for r in range(12):
            a("v_mov_b32 v%d, v%d" % (140 + r, 8 + (r + 4) % 12))   # block B's queries: another arrangement of the same random values
        for r in range(16):
            a("v_mov_b32 v%d, v%d" % (152 + r, 20 + r))
        for r in range(192, 224):
            a("v_mov_b32 v%d, 0" % r)
    a("s_mov_b32 s20, %[iters]")
    a("s_mov_b32 s21, 0")
    a("L_loop_%=:")
    KB, VB = [0, 7168], [14336, 23552]

    def qk(g, buf, sreg):
        for ks in range(3):
            a("ds_read_b128 v[%d:%d], %%[kad] offset:%d" % (52 + 4 * ks, 55 + 4 * ks, KB[buf] + g * 32 * 112 + ks * 32))
        a("s_waitcnt lgkmcnt(0)")
        for ks in range(3):
            c = "v[20:35]" if ks == 0 else "v[%d:%d]" % (sreg, sreg + 15)
            a("v_mfma_f32_32x32x16_f16 v[%d:%d], v[%d:%d], v[%d:%d], %s" % (sreg, sreg + 15, 52 + 4 * ks, 55 + 4 * ks, 8 + 4 * ks, 11 + 4 * ks, c))

    def vreads(g, buf):
        for s2 in range(2):
            for dt in range(2):
                i = s2 * 2 + dt
                a("ds_read_b128 v[%d:%d], %%[vad] offset:%d" % (72 + 4 * i, 75 + 4 * i, VB[buf] + dt * 32 * 144 + (g * 32 + 16 * s2) * 2))

    def softmax(sreg):
        a("s_nop 10")   # MFMA result -> first VALU reader (hipcc pads the product's v_med3 the same way)
        a("v_med3_f32 v6, v%d, v%d, %%[inf]" % (sreg + 14, sreg + 15))
        for r in range(0, 14, 2):
            a("v_max3_f32 v6, v6, v%d, v%d" % (sreg + r, sreg + r + 1))
        a("v_cmp_lt_f32 vcc, %[defer], v6")
        for r in range(16):
            a("v_exp_f32 v%d, v%d" % (sreg + r, sreg + r))
        for r in range(0, 16, 2):
            a("v_cvt_pkrtz_f16_f32 v%d, v%d, v%d" % (64 + r // 2, sreg + r, sreg + r + 1))
        for r in range(12):   # the rest of the group's VALU work (addresses, mask of the ragged tile, row sums)
            a("v_add_u32 v7, v7, v6")

    def pv():
        a("s_waitcnt lgkmcnt(0)")
        for s2 in range(2):
            for dt in range(2):
                i = s2 * 2 + dt
                a("v_mfma_f32_32x32x16_f16 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d]" % (88 + 16 * dt, 103 + 16 * dt, 72 + 4 * i, 75 + 4 * i, 64 + 4 * s2, 67 + 4 * s2, 88 + 16 * dt, 103 + 16 * dt))

    for buf in range(2):
        if staging:
            a("buffer_load_dwordx4 v[120:123], %[kvo], %[rsrc], s21 offen")
            a("buffer_load_dwordx4 v[124:127], %[vvo], %[rsrc], s21 offen")
            a("s_add_u32 s21, s21, 0x4000")
            a("s_and_b32 s21, s21, 0xfffff")
        if order == "product":
            for g in range(2):
                qk(g, buf, S0)
                vreads(g, buf)
                softmax(S0)
                pv()
        elif order in ("compiled", "compiled_early_v"):
            # the instruction order hipcc emits for the product kernel (common path of its steady-state loop, ISA of attn.o):
            # V^T fragments read just in time (a wait one or two instructions after the read), the exponentials of the second
            # half and the second pair of converts between the PV MFMAs, the next group's K fragments read under the PV MFMAs
            early = order == "compiled_early_v"
            def kread(g, ks):
                a("ds_read_b128 v[%d:%d], %%[kad] offset:%d" % (52 + 4 * ks, 55 + 4 * ks, KB[buf] + g * 32 * 112 + ks * 32))
            def vread(i, g, s2, dt):
                a("ds_read_b128 v[%d:%d], %%[vad] offset:%d" % (72 + 4 * i, 75 + 4 * i, VB[buf] + dt * 32 * 144 + (g * 32 + 16 * s2) * 2))
            def mf(dt, i, s2):
                a("v_mfma_f32_32x32x16_f16 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d]" % (88 + 16 * dt, 103 + 16 * dt, 72 + 4 * i, 75 + 4 * i, 64 + 4 * s2, 67 + 4 * s2, 88 + 16 * dt, 103 + 16 * dt))
            def ex(lo, hi):
                for r in range(lo, hi):
                    a("v_exp_f32 v%d, v%d" % (S0 + r, S0 + r))
            def cv(lo, hi):
                for r in range(lo, hi):
                    a("v_cvt_pk_f16_f32 v%d, v%d, v%d" % (64 + r, S0 + 2 * r, S0 + 2 * r + 1))
            for ks in range(3):
                kread(0, ks)
            a("s_waitcnt lgkmcnt(0)")
            for g in range(2):
                for ks in range(3):
                    c = "v[20:35]" if ks == 0 else "v[%d:%d]" % (S0, S0 + 15)
                    a("v_mfma_f32_32x32x16_f16 v[%d:%d], v[%d:%d], v[%d:%d], %s" % (S0, S0 + 15, 52 + 4 * ks, 55 + 4 * ks, 8 + 4 * ks, 11 + 4 * ks, c))
                if early:
                    vread(0, g, 0, 0); vread(1, g, 1, 0); vread(2, g, 0, 1); vread(3, g, 1, 1)
                a("s_nop 10")
                a("v_max_f32 v6, v%d, v%d" % (S0 + 15, S0 + 15))
                a("v_max_f32 v7, v%d, v%d" % (S0 + 14, S0 + 14))
                a("v_max_f32 v6, v7, v6")
                for r in range(0, 14, 2):
                    a("v_max3_f32 v6, v6, v%d, v%d" % (S0 + r, S0 + r + 1))
                    a("s_nop 0")
                a("v_cmp_lt_f32 vcc, %[defer], v6")
                a("s_nop 3")   # (the not-taken branch)
                a("v_add_u32 v7, v7, v6")
                a("v_add_u32 v7, v7, v6")
                a("v_add_u32 v7, v7, v6")
                ex(0, 4)
                if not early: vread(0, g, 0, 0)
                ex(4, 8)
                cv(0, 4)
                if not early: vread(1, g, 1, 0)
                if g == 0: kread(1, 2)
                ex(8, 9)
                a("s_waitcnt lgkmcnt(%d)" % (0 if early else (2 if g == 0 else 1)))
                mf(0, 0, 0)
                if not early: vread(2, g, 0, 1)
                ex(9, 12)
                if not early: vread(3, g, 1, 1)
                ex(12, 13)
                if not early: a("s_waitcnt lgkmcnt(1)")
                mf(1, 2, 0)
                ex(13, 16)
                cv(4, 8)
                a("s_nop 1")
                mf(0, 1, 1)
                if g == 0:
                    kread(1, 0); kread(1, 1)
                    a("s_waitcnt lgkmcnt(2)")
                    mf(1, 3, 1)
                    a("s_waitcnt lgkmcnt(0)")
                else:
                    a("s_waitcnt lgkmcnt(0)")
                    mf(1, 3, 1)
        elif order == "q64":
            # 64 queries per wave (two 32-query blocks A / B share every K and V^T fragment read): half the LDS reads and half the
            # staging per score, two independent MFMA chains to interleave; ~210 VGPRs = 2 waves per SIMD
            def kread(g, ks):
                a("ds_read_b128 v[%d:%d], %%[kad] offset:%d" % (52 + 4 * ks, 55 + 4 * ks, KB[buf] + g * 32 * 112 + ks * 32))
            QF = {"A": 8, "B": 140}; NM = {"A": 20, "B": 152}; SR = {"A": 36, "B": 168}; PW = {"A": 64, "B": 184}; OO = {"A": 88, "B": 192}
            def soft(blk):
                sreg, pw = SR[blk], PW[blk]
                a("v_max_f32 v6, v%d, v%d" % (sreg + 15, sreg + 15))
                a("v_max_f32 v7, v%d, v%d" % (sreg + 14, sreg + 14))
                a("v_max_f32 v6, v7, v6")
                for r in range(0, 14, 2):
                    a("v_max3_f32 v6, v6, v%d, v%d" % (sreg + r, sreg + r + 1))
                a("v_cmp_lt_f32 vcc, %[defer], v6")
                for r in range(16):
                    a("v_exp_f32 v%d, v%d" % (sreg + r, sreg + r))
                for r in range(8):
                    a("v_cvt_pk_f16_f32 v%d, v%d, v%d" % (pw + r, sreg + 2 * r, sreg + 2 * r + 1))
                for r in range(6):
                    a("v_add_u32 v7, v7, v6")
            for g in range(2):
                for ks in range(3):
                    kread(g, ks)
                a("s_waitcnt lgkmcnt(0)")
                for ks in range(3):
                    for blk in ("A", "B"):
                        c = "v[%d:%d]" % (NM[blk], NM[blk] + 15) if ks == 0 else "v[%d:%d]" % (SR[blk], SR[blk] + 15)
                        a("v_mfma_f32_32x32x16_f16 v[%d:%d], v[%d:%d], v[%d:%d], %s" % (SR[blk], SR[blk] + 15, 52 + 4 * ks, 55 + 4 * ks, QF[blk] + 4 * ks, QF[blk] + 4 * ks + 3, c))
                vreads(g, buf)
                a("s_nop 6")
                soft("A")
                soft("B")
                a("s_waitcnt lgkmcnt(0)")
                for s2 in range(2):
                    for dt in range(2):
                        i = s2 * 2 + dt
                        for blk in ("A", "B"):
                            o = OO[blk] + 16 * dt
                            a("v_mfma_f32_32x32x16_f16 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d]" % (o, o + 15, 72 + 4 * i, 75 + 4 * i, PW[blk] + 4 * s2, PW[blk] + 4 * s2 + 3, o, o + 15))
        elif order == "qk_ahead":
            qk(0, buf, S0)
            qk(1, buf, S1)
            vreads(0, buf)
            softmax(S0)
            pv()
            vreads(1, buf)
            softmax(S1)
            pv()
        if staging:
            a("s_waitcnt vmcnt(0)")
            a("ds_write_b128 %%[kwa], v[120:123] offset:%d" % KB[buf ^ 1])
            a("ds_write_b64 %%[vwa], v[124:125] offset:%d" % VB[buf ^ 1])
            a("ds_write_b64 %%[vwa], v[126:127] offset:%d" % (VB[buf ^ 1] + 16))
            a("s_waitcnt lgkmcnt(0)")
        if barrier:
            a("s_barrier")
    if skip:
        def keep(l):
            if "exp" in skip and l.startswith("v_exp_f32"):
                return l.replace("v_exp_f32", "v_mov_b32")
            if "pv" in skip and l.startswith("v_mfma") and ("v[88:103]," in l.split(",")[0] + "," or "v[104:119]," in l.split(",")[0] + ","):
                return None
            if "qk" in skip and l.startswith("v_mfma") and l.split(",")[0].endswith("v[%d:%d]" % (S0, S0 + 15)):
                return None
            return l
        lines[:] = [k for k in (keep(l) for l in lines) if k is not None]
    a("s_sub_u32 s20, s20, 1")
    a("s_cmp_lg_u32 s20, 0")
    a("s_cbranch_scc1 L_loop_%=")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a("s_nop 15")
    a("s_nop 15")
    a("v_add_f32 %[res], v88, v7")
    top = 144 if order == "qk_ahead" else (224 if order == "q64" else 128)
    clob = ["v%d" % r for r in range(6, top)] + ["s20", "s21", "s22", "scc", "vcc", "memory"]
    body = "\n".join('      "%s\\n"' % l for l in lines)
    emit("""
__global__ __launch_bounds__(512, %d) void %s(int iters, const char* src, float* sink, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  init_lds(smem, 32768);
  __syncthreads();
  const int qi = lane & 31, half = lane >> 5;
  const unsigned kad = (unsigned)(qi * 112 + half * 16), vad = (unsigned)(qi * 144 + half * 16);
  const unsigned kwa = (unsigned)((tid * 16) %% 7168), vwa = (unsigned)((tid * 16) %% 9200);
  const unsigned kvo = (unsigned)((blockIdx.x * 8192 + tid * 16) & 0xfffff), vvo = kvo + 0x100000u;
  const u32x4 rsrc = make_rsrc(src);
  const float inf = __builtin_inff(), defer = 8.0f;
  float res;
  const unsigned long long t0 = memtime(), r0 = memrealtime();
  asm volatile(
%s
      : [res] "=v"(res)
      : [kad] "v"(kad), [vad] "v"(vad), [kwa] "v"(kwa), [vwa] "v"(vwa), [kvo] "v"(kvo), [vvo] "v"(vvo), [iters] "s"(iters),
        [rsrc] "s"(rsrc), [inf] "s"(inf), [defer] "s"(defer)
      : %s);
  const unsigned long long t1 = memtime(), r1 = memrealtime();
  if (lane == 0) { atomicMax(&clk[2 * (blockIdx.x & 255)], t1 - t0); atomicMax(&clk[2 * (blockIdx.x & 255) + 1], r1 - r0); }
  if (res == 123.456f) sink[0] = res;
}
""" % (wps, name, body, ", ".join('"%s"' % c for c in clob)))
    return dict(name=name, wps=wps, qblocks=2 if order == "q64" else 1)


def mix_kernel(name, mode):
    """8 waves (2 per SIMD). mode: 'mfma', 'valu', 'both' (every wave issues the whole mix), 'split' (waves 0-3 MFMA only, waves 4-7 VALU
    only - each SIMD hosts one of each), 'both_exp' / 'both_fma': the mix with only its transcendental / only its plain VALU part."""
    NM, NE, NV = 14, 32, 60

    def seq(with_m, with_e, with_v):
        l = []
        # interleave: after each MFMA a share of the VALU work (independent registers: v40.. for exp, v80.. for fma)
        e = v = 0
        for m in range(NM):
            if with_m:
                n = m % 8
                l.append("v_mfma_f32_32x32x16_f16 a[%d:%d], v[8:11], v[12:15], a[%d:%d]" % (16 * n, 16 * n + 15, 16 * n, 16 * n + 15))
            while with_e and e < (m + 1) * NE // NM:
                l.append("v_exp_f32 v%d, v%d" % (40 + e % 32, 72 + e % 8))
                e += 1
            while with_v and v < (m + 1) * NV // NM:
                l.append("v_fma_f32 v%d, v%d, v16, v17" % (80 + v % 32, 80 + v % 32))
                v += 1
        return l

    lines = []
    a = lines.append
    for r in range(128):
        a("v_accvgpr_write_b32 a%d, 0" % r)
    for r in range(8, 16):
        a("v_mov_b32 v%d, 0x3c003c00" % r)
    a("v_mov_b32 v16, 0x3f7fff00")   # fma: x * 0.99999 + tiny
    a("v_mov_b32 v17, 0x2f800000")
    for r in range(72, 80):
        a("v_mov_b32 v%d, 0xbf000000" % r)   # exp2(-0.5): sources are constants, results independent (as the kernel's scores)
    for r in range(40, 72):
        a("v_mov_b32 v%d, 0" % r)
    for r in range(80, 112):
        a("v_mov_b32 v%d, 0x3f800000" % r)
    a("s_mov_b32 s20, %[iters]")
    if mode == "split":
        a("s_cmp_lt_u32 %[wave], 4")
        a("s_cbranch_scc0 L_valu_%=")
        a("L_m_%=:")
        for l in seq(True, False, False):
            a(l)
        a("s_sub_u32 s20, s20, 1")
        a("s_cmp_lg_u32 s20, 0")
        a("s_cbranch_scc1 L_m_%=")
        a("s_branch L_end_%=")
        a("L_valu_%=:")
        for l in seq(False, True, True):
            a(l)
        a("s_sub_u32 s20, s20, 1")
        a("s_cmp_lg_u32 s20, 0")
        a("s_cbranch_scc1 L_valu_%=")
        a("L_end_%=:")
    else:
        flags = dict(mfma=(True, False, False), valu=(False, True, True), both=(True, True, True),
                     both_exp=(True, True, False), both_fma=(True, False, True), exp=(False, True, False),
                     fma=(False, False, True))[mode]
        a("L_loop_%=:")
        for l in seq(*flags):
            a(l)
        a("s_sub_u32 s20, s20, 1")
        a("s_cmp_lg_u32 s20, 0")
        a("s_cbranch_scc1 L_loop_%=")
    a("s_nop 15")
    a("s_nop 15")
    a("v_accvgpr_read_b32 %[res], a0")
    a("v_add_f32 %[res], %[res], v40")
    a("v_add_f32 %[res], %[res], v80")
    clob = ["v%d" % r for r in range(8, 18)] + ["v%d" % r for r in range(40, 80)] + ["v%d" % r for r in range(80, 112)]
    clob += ["a%d" % r for r in range(128)] + ["s20", "scc", "memory"]
    body = "\n".join('      "%s\\n"' % l for l in lines)
    emit("""
__global__ __launch_bounds__(512) void %s(int iters, const char* src, float* sink, unsigned long long* clk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float res;
  const unsigned long long t0 = memtime(), r0 = memrealtime();
  asm volatile(
%s
      : [res] "=&v"(res)
      : [iters] "s"(iters), [wave] "s"(wave)
      : %s);
  const unsigned long long t1 = memtime(), r1 = memrealtime();
  // the slowest role decides: every wave reports, the host takes the maximum per workgroup
  if ((threadIdx.x & 63) == 0) { atomicMax(&clk[2 * blockIdx.x], t1 - t0); atomicMax(&clk[2 * blockIdx.x + 1], r1 - r0); }
  if (res == 123.456f) sink[0] = res;
}
""" % (name, body, ", ".join('"%s"' % c for c in clob)))
    return dict(name=name, mode=mode)


emit("""// GENERATED by scripts/ubench/gen_pipe_ubench.py - do not edit. See that file for what is measured and why.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned long long memrealtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ u32x4 make_rsrc(const char* p) {
  const unsigned long long a = (unsigned long long)p;
  u32x4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
  r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y);
  r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
  return r;
}
// pseudo-random fp16 values in (-1, 1): the operand bits toggle as real data would (power)
__device__ __forceinline__ void init_lds(char* smem, int bytes) {
  for (int i = threadIdx.x; i < bytes / 4; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned lo = 0x3800u | (h & 0x83ffu), hi = 0x3800u | ((h >> 16) & 0x83ffu);  // +-[0.5, 1)
    ((unsigned*)smem)[i] = lo | (hi << 16);
  }
}
""")

tiles = []
for fill in (0, 1):
    for reads in (1, 0):
        if fill and not reads:
            continue
        sfx = ("_rd" if reads else "_nord") + ("_fill" if fill else "")
        tiles.append(tile_kernel("k_t8_2x5" + sfx, 8, 2, 5, reads, fill))
        tiles.append(tile_kernel("k_t4_4x5" + sfx, 4, 4, 5, reads, fill))
        tiles.append(tile_kernel("k_t4_4x4" + sfx, 4, 4, 4, reads, fill))
        tiles.append(tile_kernel("k_t4_2x5" + sfx, 4, 2, 5, reads, fill))   # today's wave tile, one wave per SIMD
for (w, mt, nt) in ((8, 2, 5), (4, 4, 5)):
    for (rd, sync, swz) in (("spread", "none", 0), ("spread", "wait", 0), ("spread", "barrier", 0), ("late", "none", 0),
                            ("late", "barrier", 0), ("late", "barrier", 1), ("spread", "barrier", 1), ("spread", "none", 1)):
        tiles.append(kloop_kernel("k_kl%d_%dx%d_%s_%s%s" % (w, mt, nt, rd, sync, "_swz" if swz else ""), w, mt, nt, rd, sync, swz))
tiles.append(kloop_kernel("k_kl8_2x5_late_barrier_swz_sharedB", 8, 2, 5, "late", "barrier", 1, shared_b=True))
tiles.append(kloop_kernel("k_kl8_2x5_late_wait_swz_sharedB", 8, 2, 5, "late", "wait", 1, shared_b=True))
tiles.append(kloop_kernel("k_kl8_2x5_late_wait_swz", 8, 2, 5, "late", "wait", 1))
attns = [attn_kernel("k_att_product"), attn_kernel("k_att_nobarrier", barrier=False),
         attn_kernel("k_att_nostaging", staging=False), attn_kernel("k_att_nostage_nobar", staging=False, barrier=False),
         attn_kernel("k_att_product_w2", wps=2), attn_kernel("k_att_qk_ahead_w2", order="qk_ahead", wps=2),
         attn_kernel("k_att_nobarrier_w2", barrier=False, wps=2),
         attn_kernel("k_att_compiled", order="compiled"), attn_kernel("k_att_compiled_early_v", order="compiled_early_v"),
         attn_kernel("k_att_compiled_nostage", order="compiled", staging=False, barrier=False),
         attn_kernel("k_att_rand", order="compiled", data="random"),
         attn_kernel("k_att_rand_noexp", order="compiled", data="random", skip=("exp",)),
         attn_kernel("k_att_rand_nopv", order="compiled", data="random", skip=("pv",)),
         attn_kernel("k_att_rand_noqk", order="compiled", data="random", skip=("qk",)),
         attn_kernel("k_att_rand_nomfma", order="compiled", data="random", skip=("qk", "pv")),
         attn_kernel("k_att_rand_nostage", order="compiled", data="random", staging=False, barrier=False),
         attn_kernel("k_att_rand_w2", order="compiled", data="random", wps=2),
         attn_kernel("k_att_rand_q64", order="q64", data="random", wps=2),
         attn_kernel("k_att_zero_q64", order="q64", wps=2)]
mixes = [mix_kernel("k_mix_" + m, m) for m in ("mfma", "valu", "exp", "fma", "both", "both_exp", "both_fma", "split")]

emit("""
typedef void (*kern_t)(int, const char*, float*, unsigned long long*);
struct Res { double ms, ghz, cyc_per_iter; };
static Res run(kern_t k, int threads, int lds, int iters, const char* src, float* sink, unsigned long long* clk, int grid = 256) {
  HIP_OK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int reps = getenv("PIPE_REPS") ? atoi(getenv("PIPE_REPS")) : 6;  // PIPE_REPS=200: seconds of steady load (rocm-smi sampling)
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, iters, src, sink, clk);  // warm up, clocks settle
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemset(clk, 0, 512 * sizeof(unsigned long long)));
  HIP_OK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, iters, src, sink, clk);
  HIP_OK(hipEventRecord(e1));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(512);
  HIP_OK(hipMemcpy(h.data(), clk, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int b = 0; b < 256; ++b) { cyc += (double)h[2 * b]; rt += (double)h[2 * b + 1]; }
  Res r;
  r.ms = ms / reps;
  r.ghz = cyc / rt * 0.1;  // s_memrealtime: 100 MHz
  r.cyc_per_iter = cyc / 256 / iters;
  return r;
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : "";
  char* src; float* sink; unsigned long long* clk;
  HIP_OK(hipMalloc(&src, 4 << 20)); HIP_OK(hipMemset(src, 0x3a, 4 << 20));
  HIP_OK(hipMalloc(&sink, 64)); HIP_OK(hipMalloc(&clk, 512 * sizeof(unsigned long long)));
  printf("%-32s %5s %9s %7s %7s %9s %8s %6s\\n", "kernel", "waves", "mfma/iter", "reads", "fills", "TFLOP/s", "GHz", "duty");
""")
for t in tiles:
    emit("""  if (strstr("%(name)s", only)) {
    const int iters = 120000 * 20 / %(mfma_per_iter)d;
    const Res r = run(%(name)s, %(waves)d * 64, %(lds)d, iters, src, sink, clk);
    const double flop = 256.0 * %(waves)d * (double)iters * %(mfma_per_iter)d * 32768.0;
    // duty: flops over what 1024 SIMDs x 1024 flops per cycle deliver at the measured clock
    printf("%%-32s %%5d %%9d %%7d %%7d %%9.1f %%8.3f %%6.3f\\n", "%(name)s", %(waves)d, %(mfma_per_iter)d, %(reads)d, %(fill)d,
           flop / r.ms * 1e-9, r.ghz, flop / r.ms * 1e-9 / (1024.0 * 1024.0 * r.ghz * 1e-3));
  }
""" % t)
emit("""  printf("\\nattention mix per wave and 64-key tile: 14 MFMA (32 cycles each), 32 v_exp_f32, 60 v_fma_f32; 8 waves per CU\\n");
  printf("%-16s %12s %8s\\n", "kernel", "cycles/iter", "GHz");
""")
for m in mixes:
    emit("""  if (strstr("%(name)s", only)) {
    const Res r = run(%(name)s, 512, 100 * 1024, 20000, src, sink, clk);
    printf("%%-16s %%12.1f %%8.3f\\n", "%(name)s", r.cyc_per_iter, r.ghz);
  }
""" % m)
emit("""  printf("\\nattention skeleton (k_attention<48,64,40,2,false,32,8>): SIMD cycles per wave and 64-key tile (product, rocprof counters: ~1000)\\n");
  printf("%-22s %4s %12s %8s %8s   (ns per 32-query block of a wave and 64-key tile)\\n", "kernel", "w/S", "cycles", "GHz", "ns");
""")
for t in attns:
    emit("""  if (strstr("%(name)s", only)) {
    // %(wps)d waves per SIMD = %(wps)d / 2 resident 8-wave workgroups per CU, enforced through the dynamic LDS size
    const int wgs_per_cu = %(wps)d / 2;
    const int lds = wgs_per_cu == 2 ? 70 * 1024 : 100 * 1024;
    const Res r = run(%(name)s, 512, lds, 4000, src, sink, clk, 256 * wgs_per_cu);
    // a workgroup's 8 waves put 2 on every SIMD; with wgs_per_cu resident workgroups a SIMD runs 2 * wgs_per_cu wave-tiles per tile step
    printf("%%-22s %%4d %%12.1f %%8.3f %%8.1f\\n", "%(name)s", 2 * wgs_per_cu, r.cyc_per_iter / 2.0 / (2 * wgs_per_cu), r.ghz,
           r.cyc_per_iter / 2.0 / (2 * wgs_per_cu) / r.ghz / %(qblocks)d);
  }
""" % t)
emit("""  return 0;
}
""")
sys.stdout.write("\n".join(out))
