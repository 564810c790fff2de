// gram_kernels.hpp -- the sparse normal-equation kernels (gfx950 / CDNA4, wave64, MFMA 16x16x4).
//
// What the reference does per observed entry (trmf.cpp:382-389): gather one k-vector x_j and add
// the rank-1 update x_j x_j^T (upper triangle) and y*x_j.  Here four observed entries form one
// K=4 slice of an exact-f32 (or f64) MFMA:  A_tile(ti,tj) += P^T P  with P the 4 x 16 panel of
// gathered factor slices, so the per-row Gram is a dense SYRK over the gathered panel with the
// contraction running over the row's observed entries in CSR order.  v_mfma_f32_16x16x4_f32 is
// bit-for-bit an fmaf chain in k order (MI355X guide section 3), i.e. the same summation order as
// the reference's sequential loop.  The factor is stored in HBM with leading dimension
// KP = 16*NT (zero padded), so every operand load is one 64-byte segment per 16-lane group and
// needs no masking.
//
//   fsolve_quad_kernel (fp32) / fsolve_mfma_kernel (fp64)   Gram + Cholesky + substitutions -> rows of F
//                   (trmf.cpp:369-397).  Rounds 1-2's forms (one system per wavefront with one column per lane; an
//                   8 x 8 lane grid) were removed in round 5; git history and profiles/r0[1-4]_* have them.
//   gram_x_kernel   one wavefront per timestamp row: Gram + rhs of the X-side sub-problem, cached in
//                   HBM for the CG (replaces the per-Hv re-streaming of trmf.cpp:269-288).
//   loss_kernel     sum of squared residuals per timestamp row (trmf.cpp:231-245, loss part).
#pragma once

#include <type_traits>
#include <utility>

#include "common.hpp"

namespace trmf {

// ---- MFMA 16x16x4 traits ------------------------------------------------------------------------
template <typename T> struct Mfma16;
template <> struct Mfma16<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = 4*(lane>>4) + r
    static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <> struct Mfma16<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D layout differs: col = lane & 15, row = (lane>>4) + 4*r
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

// ---- cross-lane helpers -------------------------------------------------------------------------
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src_lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// 1/sqrt(p): v_rsq_f32 + one Newton step (<= ~1 ulp); f64 uses the correctly rounded forms.
__device__ __forceinline__ float inv_sqrt(float p) {
    const float r = __builtin_amdgcn_rsqf(p);
    const float h = 0.5f * p * r;
    return fmaf(r, fmaf(-h, r, 0.5f), r);
}
__device__ __forceinline__ double inv_sqrt(double p) { return 1.0 / sqrt(p); }

template <typename T> __device__ __forceinline__ T row16_allsum(T v) {   // sum over a 16-lane row
    v += __shfl_xor(v, 1, kWave);
    v += __shfl_xor(v, 2, kWave);
    v += __shfl_xor(v, 4, kWave);
    v += __shfl_xor(v, 8, kWave);
    return v;
}
template <typename T> __device__ __forceinline__ T wave_allsum(T v) {
    v = row16_allsum(v);
    v += __shfl_xor(v, 16, kWave);
    v += __shfl_xor(v, 32, kWave);
    return v;
}

template <int... Is, typename Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, Fn &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename Fn> __device__ __forceinline__ void static_for(Fn &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
// ---- Gram accumulation: a software-pipelined ring over a stream of observed entries ------------------
//
// One "group" = 4 consecutive observed entries of one row = one K=4 slice of the MFMA.  Lane
// (g = lane>>4, c = lane&15) loads entry e0 + u*estride + g and the factor slices X[j][16q + c].
// An "iteration" = D groups.  The loop body is ONE basic block with statically indexed slots, so
// the compiler's s_waitcnt vmcnt(N) are exact and the loads really stay in flight:
//     entry (idx, val)   loaded TWO iterations ahead   (slots jA / yA)
//     factor slices      loaded ONE iteration ahead    (slots x / yx)
// Nothing is selected on a freshly loaded value (a `valid ? x : 0` right after the load costs the
// full load latency every group -- measured); masked-out lanes instead point at an all-zero pad row
// of X (`zero_row`) and get y = 0 when their slot is promoted one iteration later.
// Streams are described per iteration by a wave-uniform GramDesc; rows are padded to a multiple of D
// groups (<= D-1 all-zero groups per row).
struct GramDesc {
    uint32_t e0;      // first entry of the iteration's first group
    uint32_t end;     // one past the row's last entry; end <= e0 means "no work" (all lanes masked)
    int row;          // caller-defined tag (system index); < 0 terminates the stream
};

template <typename T> __device__ __forceinline__ T row16_sum(T v) { return row16_allsum(v); }
#if defined(TRMF_F32)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_allsum_dpp(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror
    return v;
}
template <> __device__ __forceinline__ float row16_sum<float>(float v) { return row16_allsum_dpp(v); }
#endif

template <int NT> struct GramState {
    typename Mfma16<real>::acc_t acc[NT * (NT + 1) / 2];   // upper tiles, row-major over (ti<=tj)
    real b[NT];                                            // rhs partial of this lane group
    double loss;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int t = 0; t < NT * (NT + 1) / 2; t++) acc[t] = typename Mfma16<real>::acc_t{0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < NT; q++) b[q] = 0;
        loss = 0;
    }
};

template <int N> struct alignas(sizeof(real)) RealVec { real v[N]; };

// value of lane (quad base + SRC) of the caller's quad: one VALU DPP mov, no LDS
template <int SRC> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55 * SRC, 0xf, 0xf, true);
}
template <int SRC> __device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55 * SRC, 0xf, 0xf, true));
}
template <int SRC> __device__ __forceinline__ double quad_bcast(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), 0x55 * SRC, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x55 * SRC, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// RHS_PAD: y rides in the pad columns of the MFMA panel.  quad_inject<SRC>(x, y): lanes 12..15 of every
// 16-lane row (slice NT-1 there = logical columns 16(NT-1)+12 .. +15, pads when k <= 16 NT - 4) take y of
// lane (quad base + SRC), all other lanes keep x -- one DPP move with a bank mask, the same instruction
// that used to broadcast y for the rhs FMAs.  The Gram's columns KP-4..KP-1 then hold b = sum y x.
template <int SRC> __device__ __forceinline__ float quad_inject(float x, float y) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(y), 0x55 * SRC, 0xf, 0x8, false));
}
template <int SRC> __device__ __forceinline__ double quad_inject(double x, double y) {
    const long long xb = __double_as_longlong(x), yb = __double_as_longlong(y);
    const int lo = __builtin_amdgcn_update_dpp((int)(xb & 0xffffffffLL), (int)(yb & 0xffffffffLL), 0x55 * SRC, 0xf, 0x8, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(xb >> 32), (int)(yb >> 32), 0x55 * SRC, 0xf, 0x8, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int NT> __host__ __device__ constexpr bool rhs_pad_ok(int k) { return k <= kTile * NT - 8; }

// next(desc): advance a wave-uniform descriptor to the following iteration of the stream.
// row_done(row): called between iterations when the stream leaves `row` (st holds its Gram/rhs/loss).
//
// Loads per iteration (4 groups = 16 entries): ONE idx load + ONE val load (lane (g,c) fetches entry
// e0 + (c&3)*estride + g; group u then takes its entry from quad lane u with a DPP quad_perm
// broadcast) and FOUR vector loads of the gathered factor rows (column-interleaved layout: the NT
// slices of a lane are contiguous).
// The fp32 MFMA shares the vector ALUs with every other VALU instruction (scripts/ubench/mfma_valu.hip): each
// VALU cycle in this loop is a cycle the Gram does not get, so addresses are formed off the vector ALUs where
// they can be.  Entries: buffer descriptors rebuilt per iteration on the scalar unit (load_entries below).
// Gathered factor rows: one buffer descriptor over the table, address = descriptor base + ONE 32-bit lane
// offset = the row's byte offset (one multiply per iteration, promote_entries) broadcast within the quad and
// added to the lane's column bytes in a single v_add_u32_dpp per group (rounds 1-3: a DPP move + a
// multiply-add per group; before that a 64-bit multiply-add per row).  Factor tables are therefore limited to 4 GiB.
// 15 vector instructions per iteration beside the 24 MFMAs at three column tiles (27 until round 4).
// One load PER ELEMENT of the lane's slice (same descriptor, same lane offset, immediate offsets): a single
// multi-dword load would force its destination into a register tuple, and the allocator then copies the
// tuple's elements into the MFMA operand registers right after the load -- i.e. waits for it at once.
template <int NT, typename R>
__device__ __forceinline__ void load_factor_slice(R (&dst)[NT], __amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
#pragma unroll
    for (int q = 0; q < NT; q++) {
        if constexpr (sizeof(R) == 4)
            dst[q] = __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off + 4 * q, 0, 0));
        else
            dst[q] = __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b64(rsrc, byte_off + 8 * q, 0, 0));
    }
}

// TAIL (single-row streams with estride == 4 only: group u of an iteration is entries e0 + 4u .. e0 + 4u + 3): the last
// iteration of the row runs after the loop and issues the arithmetic of the groups that hold entries only.  The loop
// body stays one basic block (a branch inside it costs the exact waits); behind the loop nothing
// is in flight that a conservative wait could delay.  Skipped groups are all-zero operands: the sums are unchanged.
template <int NT, int D, bool DO_MMA, bool WITH_LOSS, bool RHS_PAD = false, bool TAIL = false, typename Next, typename RowDone>
__device__ __forceinline__ void gram_ring(GramState<NT> &st, const uint32_t *__restrict__ idx,
                                          const real *__restrict__ val, const real *__restrict__ X,
                                          uint32_t zero_row, uint32_t estride, int lane,
                                          const real (&wq)[NT], GramDesc d0, Next &&next,
                                          RowDone &&row_done) {
    static_assert(D == 4, "one quad lane per group of the iteration");
    constexpr int KP = kTile * NT;
    const uint32_t g = (uint32_t)(lane >> 4), c = (uint32_t)(lane & 15);
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<real *>(X), 0, -1 /* 4 GiB: no range check */, 0x00020000);
    const uint32_t lane_bytes = (uint32_t)NT * c * (uint32_t)sizeof(real);
    uint32_t jraw; real yraw; bool vraw;                 // entries of iteration n+2 (as loaded)
    uint32_t jsel; real ysel;                            // entries of iteration n+1 (masked)

    // Entries through buffer descriptors rebuilt per iteration ON THE SCALAR UNIT: base = the iteration's first entry, range = what
    // is left of the row, the lane's offset a loop constant -- no vector instruction forms an address, a lane past the row's end is
    // out of range and reads 0 (so y needs no mask).  (Rounds 1-3: add, compare, clamp, a 64-bit shift and two 64-bit adds per
    // iteration on the vector ALUs, which the fp32 MFMAs share.)
    const uint32_t lane_e = (c & 3u) * estride + g;      // the lane's entry of an iteration
    auto load_entries = [&](const GramDesc &d) {
        // wave-uniform; readfirstlane pins it (and with it the descriptors) to the scalar unit: written as a select the compiler
        // forms a saturating VECTOR subtract and then walks the "divergent" descriptor in a waterfall loop
        const uint32_t e0u = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.e0), endu = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.end);
        uint32_t left;                                   // max(end, e0) - e0 on the scalar unit (see above)
        asm("s_max_u32 %0, %1, %2\n\ts_sub_u32 %0, %0, %2" : "=&s"(left) : "s"(endu), "s"(e0u) : "scc");
        vraw = lane_e < left;
        const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(idx + e0u), 0, (int)(left * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<real *>(val + e0u), 0, (int)(left * (uint32_t)sizeof(real)), 0x00020000);
        jraw = __builtin_amdgcn_raw_buffer_load_b32(ir, lane_e * 4u, 0, 0);
        real yv[1];
        load_factor_slice(yv, vr, lane_e * (uint32_t)sizeof(real));
        yraw = yv[0];
    };
    auto promote_entries = [&]() {                       // consumes loads issued one iteration ago
        // the factor row's BYTE offset (rows < 2^24, checked at session creation): multiplied once per iteration here, so that a
        // group's address is one add whose first operand is the DPP quad broadcast (v_add_u32_dpp) instead of a move + multiply-add
        jsel = __umul24(vraw ? jraw : zero_row, (uint32_t)(KP * sizeof(real)));
        ysel = yraw;                                     // 0 past the row's end (out of the descriptor's range)
    };
    auto load_slices = [&](auto U, real (&x)[D][NT], real (&yx)[D]) {   // slot u <- factor row of group u
        constexpr int u = decltype(U)::value;
        if constexpr (!RHS_PAD) yx[u] = quad_bcast<u>(ysel);
        load_factor_slice(x[u], x_rsrc, quad_bcast<u>(jsel) + lane_bytes);
    };

    GramDesc d1 = d0; next(d1);
    GramDesc d2 = d1; next(d2);
    real x[D][NT], yx[D];
    load_entries(d0);
    promote_entries();
    // entries of iteration 1 BEFORE the slices of iteration 0, the order of the loop body: the loop's first wait (for the
    // entries loaded one iteration ago) is then "all but the four slice loads behind them" on both paths into the loop
    // header.  With the entries requested last here (rounds 1-3) the wait merged over both paths was vmcnt(0): every
    // iteration drained the four gathers it had just issued, and a wavefront's ring only overlapped with OTHER wavefronts.
    load_entries(d1);
    static_for<D>([&](auto U) { load_slices(U, x, yx); });
    // Slot u is consumed (its MFMAs and rhs FMAs issued) and then immediately re-requested for the next
    // iteration, in place: every load still has a full iteration of MFMA time (24 x 32 cycles) to land, and
    // there is no second operand set to copy into (12 + 4 register moves per iteration that the shared
    // f32 ALUs would have to execute, section 4.1 (3) of DESIGN.md).
    auto consume = [&](auto U, real ycons) {             // the arithmetic of group u of the iteration being consumed
        constexpr int u = decltype(U)::value;
        real xt[NT];                                     // operands of group u; the last slice carries y in its pads
#pragma unroll
        for (int q = 0; q < NT; q++) xt[q] = x[u][q];
        if constexpr (RHS_PAD) xt[NT - 1] = quad_inject<u>(x[u][NT - 1], ycons);
        else {
#pragma unroll
            for (int q = 0; q < NT; q++) st.b[q] = fma(yx[u], x[u][q], st.b[q]);
        }
        if (DO_MMA) {
            int t = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ti++)
#pragma unroll
                for (int tj = ti; tj < NT; tj++, t++) st.acc[t] = Mfma16<real>::mma(xt[ti], xt[tj], st.acc[t]);
        }
        if (WITH_LOSS) {
            real d = 0;
#pragma unroll
            for (int q = 0; q < NT; q++) d = fma(wq[q], x[u][q], d);
            d = row16_sum(d);
            const real res = yx[u] - d;                   // trmf.cpp:238 (val_type arithmetic)
            st.loss += (double)res * (double)res;         // masked lanes: y = 0, x = 0 -> 0
        }
    };
    while (TAIL ? d1.row >= 0 : d0.row >= 0) {
        // ---- single basic block ----
        const real ycons = ysel;                         // y of the iteration being consumed (RHS_PAD)
        promote_entries();                               // entries of n+1 (loaded one iteration ago)
        load_entries(d2);                                // entries of n+2
        static_for<D>([&](auto U) {
            consume(U, ycons);
            // pin the order "all of group u's arithmetic, then its reload": left alone, the scheduler moves the
            // reload above MFMAs that still read the slot, renames it and waits for the fresh load at once
            __builtin_amdgcn_sched_barrier(0);
            load_slices(U, x, yx);                       // slot u <- group u of iteration n+1
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- between iterations ----
        if (d1.row != d0.row) { row_done(d0.row); }
        d0 = d1; d1 = d2; next(d2);
    }
    if constexpr (TAIL) {
        if (d0.row >= 0) {                               // the row's last iteration: groups with entries only
            const int ngr = (int)((d0.end - d0.e0 + 3u) >> 2);
            const real ycons = ysel;
            static_for<D>([&](auto U) {
                if (decltype(U)::value < ngr) consume(U, ycons);
            });
            row_done(d0.row);
        }
    }
}

// Single-row stream: groups e0, e0+estride, ... of [p0, p1).
struct SingleRowStream {
    uint32_t step;    // entries per iteration = D * estride
    __device__ __forceinline__ void operator()(GramDesc &d) const {
        if (d.row < 0) return;
        d.e0 += step;
        if (d.e0 >= d.end) { d.row = -1; d.end = 0; }
    }
};

constexpr int kRingDepth = 4;      // groups per iteration of gram_ring (rows padded to a multiple)

// ---- split rows (round 6): rows too long for the static row -> wavefront mapping ---------------------------------------------
// The reference schedules rows dynamically (trmf.cpp:371 `schedule(dynamic,64)`, :234,252,273 `schedule(dynamic,32)`) and is
// indifferent to a row's length; here a row's entries are streamed by ONE wavefront, so a long row -- a densely observed series
// (the paper's data used for imputation: 370 rows of ~21 000 entries), a hub item or a fully observed timestamp of a power-law
// pattern -- is a serial chain that the rest of the chip waits for.  Rows of at least `thresh` entries (session_state.hpp:
// LongRows, decided per orientation at set-up) therefore leave the row kernels (which skip them) and are cut into ITEMS of
// contiguous entries:
//   gram_part_kernel      one wavefront per item: gram_ring over the item's entries -> a partial Gram (+ rhs) in the MFMA
//                         accumulator layout, stored to a slab (16 B per lane and tile: whole cache lines);
//   fsolve_*_long_kernel / gram_x_long_kernel
//                         the row kernels' second halves (+lambda, factorisation, substitutions -> the row of F; G_i / b_i of
//                         a timestamp) fed by the sum of the row's partials IN ITEM ORDER -- a fixed order, so the result does
//                         not depend on the launch geometry or the rank count.
// Within an item the contraction runs in CSR order like everywhere else; across items the partial sums are added left to right.
// That is a different rounding from the reference's single chain (by about an ulp of the row's Gram per item); rows below the
// threshold are computed exactly as before.
struct SplitRows {
    const uint32_t *rows;      // ids of the long rows, ascending
    const uint32_t *first;     // first item of long row i (i = 0 .. count: one past the last at the end)
    const real *slab;          // partial Grams, `stride` reals per item
    uint32_t stride;
    uint32_t begin, end;       // positions of the long-row list this launch covers
};
template <int NT> __host__ __device__ constexpr uint32_t split_part_reals(bool with_b) {
    return (uint32_t)(NT * (NT + 1) / 2) * 4u * kWave + (with_b ? (uint32_t)NT * kWave : 0u);
}
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
// The sum of a split row's partials, left to right, INTO its first item's slot: one thread per 16-byte granule of a partial, the
// items' loads in flight eight at a time.  (First form of this round: the consumer kernels summed the partials themselves, one
// wavefront per row / per four rows with a whole partial per load round -- 158 us for the 370 x 21 partials of the `imp` workload,
// a chain of memory round trips; this kernel: see profiles/r06_split_rows.txt.)  Same additions in the same order.
__global__ __launch_bounds__(256) void split_reduce_kernel(const uint32_t *__restrict__ first, real *__restrict__ slab, uint32_t stride,
                                                           uint32_t begin, uint32_t end, uint32_t blocks_per_row) {
    typedef real gran_t __attribute__((ext_vector_type(16 / sizeof(real))));
    constexpr uint32_t GR = 16 / sizeof(real);
    const uint32_t li = begin + blockIdx.x / blocks_per_row;
    if (li >= end) return;
    const uint32_t i0 = first[li], i1 = first[li + 1];
    const uint32_t gidx = (blockIdx.x % blocks_per_row) * 256u + threadIdx.x;
    if (i1 - i0 < 2u || gidx * GR >= stride) return;
    real *dst = slab + (size_t)i0 * stride + (size_t)gidx * GR;
    gran_t acc = *reinterpret_cast<const gran_t *>(dst);
    const real *p = dst + stride;
    uint32_t it = i0 + 1;
    for (; it + 8 <= i1; it += 8, p += (size_t)8 * stride) {
        gran_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const gran_t *>(p + (size_t)u * stride);
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    for (; it < i1; it++, p += stride) acc += *reinterpret_cast<const gran_t *>(p);
    *reinterpret_cast<gran_t *>(dst) = acc;
}
#endif

// st <- the summed partial of a split row (slot of its first item, after split_reduce_kernel)
template <int NT, bool WITH_B>
__device__ __forceinline__ void load_row_sum(GramState<NT> &st, const SplitRows &sp, uint32_t i0, int lane) {
    typedef typename Mfma16<real>::acc_t acc_t;
    constexpr int NTT = NT * (NT + 1) / 2;
    const real *p = sp.slab + (size_t)i0 * sp.stride;
#pragma unroll
    for (int t = 0; t < NTT; t++) st.acc[t] = *reinterpret_cast<const acc_t *>(p + t * 4 * kWave + 4 * lane);
#pragma unroll
    for (int q = 0; q < NT; q++) st.b[q] = WITH_B ? p[NTT * 4 * kWave + q * kWave + lane] : real(0);
    st.loss = 0;
}

#if !defined(TRMF_F32)
__device__ __forceinline__ void wave_lds_sync() {       // LDS operations of one wavefront retire in order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#endif



// ---- F-solve, fp64, factorisation IN the accumulator layout (round 3; config 5's kernel) --------------------------------
// Round 2's kernel moved the Gram out of the MFMA accumulators (LDS staging into an 8 x 8 lane grid) and eliminated
// one pivot at a time: 64 LDS round trips, ~1250 + 290 FMAs, 350 multiplies and 950 LDS instructions per rank-64
// system.  Here the k x k system never leaves the registers the Gram was accumulated in:
//
//   * v_mfma_f64_16x16x4_f64 keeps row (lane>>4) + 4 r, column lane & 15 of a 16 x 16 tile in register r.  FOUR
//     CONSECUTIVE rows 4 r0 .. 4 r0 + 3 of a tile row are therefore register r0 of the four 16-lane groups -- which is
//     exactly the K = 4 operand layout of the same instruction (lane group = contraction index).  A panel of four
//     pivot rows R'_p (after the elimination inside the panel) updates every trailing tile with ONE MFMA,
//         A22[16 ti + m][16 tj + n] -= sum_p (R'_p[16 ti + m] / d_p) R'_p[16 tj + n],
//     whose B operand IS the pivot-row register of tile column tj and whose A operand is the same register of tile
//     column ti scaled by -1/d_p: no data movement at all.  80 MFMAs replace the ~1250 FMAs + their broadcasts.
//   * Inside a panel (4 rows x <= 64 columns) the four lane groups hold one row each; the rows are transposed through a
//     small LDS record per column (80-byte pitch: conflict-free 16-byte accesses) so that lane l owns COLUMN l with all
//     four rows: the 4 x 4 diagonal block is read by every lane (broadcast) and factorised redundantly (L D L^T,
//     v_rcp_f64 + Newton), each lane eliminates its column (6 FMAs), forms the scaled copy, and the lane groups read
//     their row back as the two MFMA operands.  One write + one read per tile column each way.
//   * The right-hand side lives one element per lane (lane l = row l) and follows with four FMAs per panel (the
//     multipliers of its row are the scaled column the lane has just computed); lane j0 -- whose own column is the
//     panel's first pivot column and needs no work -- eliminates the panel's four rhs entries.
//   * Back substitution, column oriented, in the same lane = row layout: the strict upper triangle of one tile column
//     at a time is staged column-major in LDS (rows at and below the diagonal as exact zeros, so finished lanes are
//     never touched and the solution is simply y / d at the end); a step is multiply, v_readlane, one conflict-free LDS
//     read and one FMA.
// posv('U') of the reference (rf_matrix.h:3008-3014) and this differ by rounding only (fp64 gate 1e-6, tests/).
#if !defined(TRMF_F32)
#ifndef TRMF_MFMA_WAVES
#define TRMF_MFMA_WAVES 2
#endif
#ifndef TRMF_RCP_NEWTON
#define TRMF_RCP_NEWTON 2
#endif
__device__ __forceinline__ double recip_pivot(double d) {
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int i = 0; i < TRMF_RCP_NEWTON; i++) { const double e = fma(-d, r, 1.0); r = fma(r, e, r); }
    return r;
}
__host__ __device__ constexpr int upper_tile_index(int ti, int tj, int NT) { return ti * NT - ti * (ti - 1) / 2 + (tj - ti); }

// acc += sum_p (lane BC of the caller's 16-lane row: S.v[p]) * R.v[p] -- four v_fmac_f64 with a DPP row broadcast on the first
// factor.  Inline assembly: the compiler keeps the broadcast as a separate v_mov_b64_dpp (one more fp64-rate instruction
// per product).  The leading s_nop covers the two wait states a DPP read needs after a VALU write of its source.
template <int BC>
__device__ __forceinline__ void fmac4_row_bcast(double &acc, const Quad<double> &S, const Quad<double> &R) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %1, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %2, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %3, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(S.v[0]), "v"(S.v[1]), "v"(S.v[2]), "v"(S.v[3]), "v"(R.v[0]), "v"(R.v[1]), "v"(R.v[2]), "v"(R.v[3]), "n"(BC));
}
#ifndef TRMF_MFMA_TAIL
#define TRMF_MFMA_TAIL 1       // the Gram's last iteration issues its non-empty groups only (gram_ring TAIL): config 5 rows have ~50 entries
#endif
#ifndef TRMF_TRAIL_MFMA
#define TRMF_TRAIL_MFMA 1      // measured at config 5: 12.6-12.8 ms with the MFMA trailing update, 13.0 ms with the DPP-fused FMAs (7b)
#endif

#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int NT, int KMAX, bool TRAIL_MFMA = (TRMF_TRAIL_MFMA != 0)>
__global__ void fsolve_mfma_kernel(const uint32_t *__restrict__ ptr,
                                                                          const uint32_t *__restrict__ idx,
                                                                          const real *__restrict__ val,
                                                                          const real *__restrict__ X,
                                                                          real *__restrict__ F, uint32_t row_begin,
                                                                          uint32_t row_end, int k, real lambda,
                                                                          uint32_t zero_row, uint32_t long_m1);
template <int NT, int KMAX, bool TRAIL_MFMA = (TRMF_TRAIL_MFMA != 0)>
__global__ void fsolve_mfma_long_kernel(SplitRows sp, real *__restrict__ F, int k, real lambda);
#else
// LONG: the system of a split row -- one wavefront per long row, its Gram and right-hand side are the sum of the row's partials
template <int NT, int KMAX, bool TRAIL_MFMA, bool LONG>
__device__ __forceinline__ void fsolve_mfma_body(const uint32_t *__restrict__ ptr,
                                                                          const uint32_t *__restrict__ idx,
                                                                          const real *__restrict__ val,
                                                                          const real *__restrict__ X,
                                                                          real *__restrict__ F, uint32_t row_begin,
                                                                          uint32_t row_end, int k, real lambda,
                                                                          uint32_t zero_row, uint32_t long_m1, const SplitRows &sp) {
    static_assert(sizeof(real) == 8, "the in-accumulator F-solve is the fp64 path");
    constexpr int KP = kTile * NT, NPAN = KMAX / 4;
    constexpr int CP = 10;                                // doubles per column record: R'0 R'1 R'2 R'3 | S0 S1 S2 S3 | 2 pad (80-byte pitch)
    constexpr int FIN = (KP + 1) * CP;                    // after the records of columns 0..KP-1 and of the rhs: (z'_q, 1/d_q) x 4
    constexpr int RP = KP + 2;                            // row pitch of the back substitution's column buffer
    constexpr int SCR = (FIN + 8 > kTile * RP) ? FIN + 8 : kTile * RP;
    __shared__ __attribute__((aligned(16))) real scr_s[4][SCR];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    real *ps = scr_s[wave];
    typedef typename Mfma16<real>::acc_t acc_t;
    typedef VecOf<real, 2> pair_t;
    GramState<NT> st;
    uint32_t row;
    if constexpr (LONG) {
        const uint32_t li = sp.begin + blockIdx.x * 4u + (uint32_t)wave;
        if (li >= sp.end) return;                       // wave-uniform; no block barrier below
        row = (uint32_t)__builtin_amdgcn_readfirstlane((int)sp.rows[li]);
        load_row_sum<NT, true>(st, sp, (uint32_t)__builtin_amdgcn_readfirstlane((int)sp.first[li]), lane);
    } else {
        row = row_begin + blockIdx.x * 4u + (uint32_t)wave;
        if (row >= row_end) return;                     // wave-uniform; no block barrier below
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ptr[row]);
        const uint32_t p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ptr[row + 1]);
        if (p1 - p0 - 1u >= long_m1) return;            // trmf.cpp:374: empty rows stay untouched; long rows: fsolve_mfma_long_kernel
        st.clear();
        real nowq[NT];
#pragma unroll
        for (int q = 0; q < NT; q++) nowq[q] = 0;
        gram_ring<NT, kRingDepth, true, false, false, TRMF_MFMA_TAIL != 0>(st, idx, val, X, zero_row, 4u, lane, nowq, GramDesc{p0, p1, 0},
                                                                          SingleRowStream{4u * kRingDepth}, [](int) {});
    }
    // right-hand side: fold the 4 lane groups, then lane l = 16 q + c keeps b[l]
    real y = 0, dinv = 1;
#pragma unroll
    for (int q = 0; q < NT; q++) {
        real v = st.b[q];
        v += __shfl_xor(v, 16, kWave);
        v += __shfl_xor(v, 32, kWave);
        if (g == q) y = v;
    }
    // + lambda on the diagonal (trmf.cpp:393); pad rows get a unit diagonal: their steps are no-ops
#pragma unroll
    for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (c == g + 4 * r) st.acc[upper_tile_index(ti, ti, NT)][r] += (kTile * ti + c < k) ? lambda : real(1);

    // ---- elimination, four pivots per step ----
    static_for<NPAN>([&](auto Pn) {
        constexpr int pn = decltype(Pn)::value, tp = pn >> 2, r0 = pn & 3, j0 = 4 * pn;
        // (1) the panel's rows (lane group g holds row j0 + g) -> column records; the four rhs entries -> record KP
#pragma unroll
        for (int tj = tp; tj < NT; tj++) ps[(kTile * tj + c) * CP + g] = st.acc[upper_tile_index(tp, tj, NT)][r0];
        if ((unsigned)(lane - j0) < 4u) ps[KP * CP + (lane - j0)] = y;
        wave_lds_sync();
        // (2) the 4 x 4 diagonal block (same addresses in every lane: broadcast reads), (3) its L D L^T
        const real *pc = ps + j0 * CP;
        const real P00 = pc[0];
        real P01 = pc[CP], P11 = pc[CP + 1];
        real P02 = pc[2 * CP], P12 = pc[2 * CP + 1], P22 = pc[2 * CP + 2];
        real P03 = pc[3 * CP], P13 = pc[3 * CP + 1], P23 = pc[3 * CP + 2], P33 = pc[3 * CP + 3];
        const real i0 = recip_pivot(P00);
        const real l10 = P01 * i0, l20 = P02 * i0, l30 = P03 * i0;
        P11 = fma(-l10, P01, P11); P12 = fma(-l10, P02, P12); P13 = fma(-l10, P03, P13);
        P22 = fma(-l20, P02, P22); P23 = fma(-l20, P03, P23); P33 = fma(-l30, P03, P33);
        const real i1 = recip_pivot(P11);
        const real l21 = P12 * i1, l31 = P13 * i1;
        P22 = fma(-l21, P12, P22); P23 = fma(-l21, P13, P23); P33 = fma(-l31, P13, P33);
        const real i2 = recip_pivot(P22);
        const real l32 = P23 * i2;
        P33 = fma(-l32, P23, P33);
        const real i3 = recip_pivot(P33);
        // (4) lane l owns column l (lane j0: the rhs column -- its own column is the first pivot column and needs nothing)
        const bool has_col = KP == kWave || lane < KP;
        const int col = (lane == j0 || !has_col) ? KP : lane;
        real *rec = ps + col * CP;
        const Quad<real> rin = *reinterpret_cast<const Quad<real> *>(rec);
        const real R0 = rin.v[0];
        const real R1 = fma(-l10, R0, rin.v[1]);
        const real R2 = fma(-l21, R1, fma(-l20, R0, rin.v[2]));
        const real R3 = fma(-l32, R2, fma(-l31, R1, fma(-l30, R0, rin.v[3])));
        const real S0 = -(R0 * i0), S1 = -(R1 * i1), S2 = -(R2 * i2), S3 = -(R3 * i3);
        if (has_col) {
            *reinterpret_cast<Quad<real> *>(rec) = Quad<real>{{R0, R1, R2, R3}};
            *reinterpret_cast<Quad<real> *>(rec + 4) = Quad<real>{{S0, S1, S2, S3}};
        }
        if (lane == j0) {                                // eliminated rhs entries z'_q and the pivots' reciprocals
            *reinterpret_cast<Quad<real> *>(ps + FIN) = Quad<real>{{R0, i0, R1, i1}};
            *reinterpret_cast<Quad<real> *>(ps + FIN + 4) = Quad<real>{{R2, i2, R3, i3}};
        }
        wave_lds_sync();
        // (5) the rhs follows: rows below the panel take four FMAs with the multipliers of their row (the scaled column
        //     the lane has just formed), the panel's rows take their final values
        {
            const Quad<real> f0 = *reinterpret_cast<const Quad<real> *>(ps + FIN), f1 = *reinterpret_cast<const Quad<real> *>(ps + FIN + 4);
            const real upd = fma(S3, f1.v[2], fma(S2, f1.v[0], fma(S1, f0.v[2], fma(S0, f0.v[0], y))));
            const int q = min(max(lane - j0, 0), 3);
            const pair_t mine = *reinterpret_cast<const pair_t *>(ps + FIN + 2 * q);
            const bool in_panel = (unsigned)(lane - j0) < 4u;
            y = in_panel ? mine.v[0] : (lane > j0 + 3 ? upd : y);
            dinv = in_panel ? mine.v[1] : dinv;
        }
        // (6) lane group g reads its row back (the final pivot row stays in the accumulators for the back substitution)
#pragma unroll
        for (int tj = tp; tj < NT; tj++) st.acc[upper_tile_index(tp, tj, NT)][r0] = ps[(kTile * tj + c) * CP + g];
        if constexpr (TRAIL_MFMA) {
            // (7a) trailing update on the matrix pipe: the scaled copy of the lane group's row is the A operand, the pivot
            //      row itself the B operand of ONE v_mfma_f64_16x16x4 per trailing tile
            real aop[NT];
#pragma unroll
            for (int tj = tp; tj < NT; tj++) aop[tj] = ps[(kTile * tj + c) * CP + 4 + g];
            if (c <= 4 * r0 + 3) aop[tp] = 0;            // rows at and above the panel are final: the update leaves them alone
            wave_lds_sync();
#pragma unroll
            for (int ti = tp; ti < NT; ti++)
#pragma unroll
                for (int tj = ti; tj < NT; tj++) {
                    acc_t &C = st.acc[upper_tile_index(ti, tj, NT)];
                    C = Mfma16<real>::mma(aop[ti], st.acc[upper_tile_index(tp, tj, NT)][r0], C);
                }
        } else {
            // (7b) trailing update on the vector ALUs.  v_mfma_f64_16x16x4 issues in ~100 cycles per SIMD on gfx950
            //      (scripts/ubench/f64_pipe.hip: 20 flop/clk against 26-32 of v_fma_f64), and a whole-tile update cannot skip
            //      the rows of a tile row that are already final; here register r of tile (ti, tj) takes
            //          A[16 ti + g + 4 r][16 tj + c] += sum_p S_p[16 ti + g + 4 r] R'_p[16 tj + c]
            //      as four v_fmac_f64 whose row operand is a DPP row broadcast (row_newbcast is the one DPP mode fp64 has): the
            //      lane reads the scaled rows of the column its 16-lane row is ROTATED to, column 16 ti + (c + g) mod 16, so
            //      that lane 4 r of every row holds S_p[16 ti + g + 4 r].  Rows at and above the panel (registers r <= r0 of
            //      tile row tp) are skipped: 880 FMAs per rank-64 system, no masks, no MFMA hazard nops.
            Quad<real> Rq[NT];
#pragma unroll
            for (int tj = tp; tj < NT; tj++) Rq[tj] = *reinterpret_cast<const Quad<real> *>(ps + (kTile * tj + c) * CP);
            static_for<NT>([&](auto Ti) {
                constexpr int ti = decltype(Ti)::value;
                if constexpr (ti >= tp) {
                    const Quad<real> Sq = *reinterpret_cast<const Quad<real> *>(ps + (kTile * ti + ((c + g) & 15)) * CP + 4);
                    static_for<4>([&](auto Rr) {
                        constexpr int r = decltype(Rr)::value;
                        if constexpr (ti > tp || r > r0) {
#pragma unroll
                            for (int tj = ti; tj < NT; tj++) {
                                real a = st.acc[upper_tile_index(ti, tj, NT)][r];
                                fmac4_row_bcast<4 * r>(a, Sq, Rq[tj]);
                                st.acc[upper_tile_index(ti, tj, NT)][r] = a;
                            }
                        }
                    });
                }
            });
            wave_lds_sync();
        }
    });

    // ---- back substitution (D L^T) x = z', column oriented, lane = row ----
    static_for<NT>([&](auto Tr) {
        constexpr int tt = NT - 1 - decltype(Tr)::value;
        if constexpr (kTile * tt < KMAX) {
            // strict upper triangle of tile column tt, column-major: cb[n * RP + row]; rows at / below the diagonal and the
            // sixteen rows beyond this tile column as exact zeros (lanes whose unknown is final are never touched)
#pragma unroll
            for (int ti = 0; ti <= tt; ti++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    real v = st.acc[upper_tile_index(ti, tt, NT)][r];
                    if (ti == tt && g + 4 * r >= c) v = 0;
                    ps[c * RP + kTile * ti + g + 4 * r] = v;
                }
            if constexpr (tt + 1 < NT) {
#pragma unroll
                for (int r = 0; r < 4; r++) ps[c * RP + kTile * (tt + 1) + g + 4 * r] = 0;
            }
            wave_lds_sync();
            static_for<kTile>([&](auto Nr) {
                constexpr int n = kTile - 1 - decltype(Nr)::value, t = kTile * tt + n;
                if constexpr (t < KMAX && t > 0) {
                    const real xt = lane_bcast(y * dinv, t);
                    const real cv = (KP == kWave || lane < KP) ? ps[n * RP + lane] : real(0);
                    y = fma(-cv, xt, y);
                }
            });
            wave_lds_sync();
        }
    });
    if (lane < k) F[(size_t)row * KP + colpos(lane, NT)] = y * dinv;
}
template <int NT, int KMAX, bool TRAIL_MFMA = (TRMF_TRAIL_MFMA != 0)>
__global__ __launch_bounds__(256, TRMF_MFMA_WAVES) void fsolve_mfma_kernel(const uint32_t *__restrict__ ptr,
                                                                          const uint32_t *__restrict__ idx,
                                                                          const real *__restrict__ val,
                                                                          const real *__restrict__ X,
                                                                          real *__restrict__ F, uint32_t row_begin,
                                                                          uint32_t row_end, int k, real lambda,
                                                                          uint32_t zero_row, uint32_t long_m1) {
    fsolve_mfma_body<NT, KMAX, TRAIL_MFMA, false>(ptr, idx, val, X, F, row_begin, row_end, k, lambda, zero_row, long_m1, SplitRows{});
}
template <int NT, int KMAX, bool TRAIL_MFMA = (TRMF_TRAIL_MFMA != 0)>
__global__ __launch_bounds__(256, TRMF_MFMA_WAVES) void fsolve_mfma_long_kernel(SplitRows sp, real *__restrict__ F, int k, real lambda) {
    fsolve_mfma_body<NT, KMAX, TRAIL_MFMA, true>(nullptr, nullptr, nullptr, nullptr, F, 0u, 0u, k, lambda, 0u, 0u, sp);
}
#endif
#endif  // !TRMF_F32

#if defined(TRMF_F32)
// ---- F-solve, quad form (fp32): one wavefront per FOUR item rows ------------------------------------
// The O(k^3) part of the solve is a chain of rank-1 updates whose operands must be broadcast across
// lanes.  With one system per wavefront (round 1) every broadcast is a v_readlane that
// feeds a single FMA per lane, and the kernel is VALU-issue bound (~3000 of its ~3500 instructions).
// Here the four 16-lane rows of a wavefront each own one system: lane (grp, c) holds columns
// {c, 16+c, 32+c, ...} of system `grp`, one ds_swizzle row-broadcast serves all four systems and
// feeds up to NT FMAs per lane, and the back substitution reduces inside the 16-lane row with DPP.
// The four Grams are accumulated one after the other by ONE gram_ring() stream that runs across the
// row boundaries (the rows are adjacent in CSR), so the gather pipeline never drains inside a quad.

// value of lane SRC (0..15) of the caller's own 16-lane row.  Round 6: a DPP move (v_mov_b32_dpp row_newbcast:SRC, a vector-ALU
// instruction with a few cycles of latency) instead of ds_swizzle, which travels through the LDS pipe (~100 cycles): the pivot
// broadcast heads the dependent chain of every elimination step (broadcast -> rsq -> scale -> quad broadcast -> MFMA), 40 steps per
// system, and with three wavefronts per SIMD that latency is not hidden (profiles/r06_fsolve_chain.txt).  Same value either way.
#ifndef TRMF_ROW_BCAST_DPP
#define TRMF_ROW_BCAST_DPP 1
#endif
template <int SRC> __device__ __forceinline__ float row_bcast(float v) {
#if TRMF_ROW_BCAST_DPP
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + SRC, 0xf, 0xf, true));
#else
    constexpr int pattern = (SRC << 5) | 0x10;          // and_mask = 0x10, or_mask = SRC, xor_mask = 0
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), pattern));
#endif
}

template <int NT> __device__ __forceinline__ constexpr int quad_slab_floats() {
    // column-major slab holding only the upper tiles: column 16q+c keeps rows 0..16q+15 (+4 pad)
    int n = 0;
    for (int q = 0; q < NT; q++) n += kTile * (kTile * (q + 1) + 4);
    return n;
}
template <int NT> __device__ __forceinline__ constexpr int quad_slab_col_offset(int q) {
    int n = 0;
    for (int i = 0; i < q; i++) n += kTile * (kTile * (i + 1) + 4);
    return n;
}

// Four right-looking Cholesky factorisations side by side (one per 16-lane row), then the row-oriented back
// substitution.  a4[q][R][i] = A[4R+i][16q+c] (+lambda on the diagonal, unit diagonal on pad rows so that no
// k-guard is needed); returns x[q] = x[16q+c].
//
// Step j scales row j (u = A[j][.] / sqrt(A[j][j]), zero left of and on the diagonal) and removes its outer
// product from the rows below.  That rank-1 update runs on the matrix pipe, four rows at a time:
// v_mfma_f32_4x4x1_16b_f32 is 16 independent 4x4 outer products -- block = 4 neighbouring lanes, result VGPR i
// of lane 4b+jj = C + A[lane 4b+i] * B[lane 4b+jj] -- which in this layout (lane = column, VGPR = row) is exactly
//     A[4R+i][16q+c] -= u[4R+i] * u[16q+c]      for i = 0..3, all 16 columns c of block q, all four systems,
// with the B operand the lane's own u[q] and the A operand one ds_swizzle (the quad holding u[4R..4R+3]
// replicated to every quad of the 16-lane row).  One swizzle + (NT - q(R)) MFMAs per row quad replace four
// swizzles + 4 (NT - q) FMAs, and an MFMA issues 512 flops in 8 cycles where four v_fma_f32 take 16
// (scripts/ubench/mfma4x4.hip); the product is rounded once like fmaf, so the arithmetic is unchanged.
// Rows <= j inside the first quad see u = 0 and stay as they are.
//
// RHS_IN: the right-hand side sits in column KP-1 of the matrix (the pad column the Gram MFMAs accumulated
// sum(y x) into, gram_ring RHS_PAD).  The forward substitution then needs no code at all -- column KP-1 is
// scaled and updated like every other column, and afterwards holds z = U^-T b -- and the back substitution
// treats it as one more unknown fixed at -1:  x_j = -(sum_{t>j} U[j][t] x_t) / U[j][j]  with  x_{KP-1} = -1.
typedef float quad_f4 __attribute__((ext_vector_type(4)));

// value of the quad Q (lanes 4Q..4Q+3) of the caller's 16-lane row, replicated to all four quads of that row
template <int Q> __device__ __forceinline__ float row_quad_bcast(float v) {
    constexpr int pattern = ((Q << 2) << 5) | 0x13;      // and_mask = 0b10011, or_mask = 4Q, xor_mask = 0
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), pattern));
}

template <int NT, int KMAX, bool RHS_IN, int ABL = 0>
__device__ __forceinline__ void quad_factor_solve(quad_f4 (&a4)[NT][KMAX / 4], float (&bz)[NT], float (&x)[NT], int c) {
    constexpr int NR = KMAX / 4;                         // row quads
    float dinv[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) dinv[q] = 0;
    if constexpr (!(ABL & 2))
    static_for<KMAX>([&](auto J) {
        constexpr int j = decltype(J)::value, qj = j >> 4, cj = j & 15, Rj = j >> 2, ij = j & 3;
        {   // one straight-line block: the scheduler can software-pipeline across steps
            const float inv = __builtin_amdgcn_rsqf(row_bcast<cj>(a4[qj][Rj][ij]));
            float u[NT], nu[NT];
#pragma unroll
            for (int q = qj; q < NT; q++) {
                u[q] = a4[q][Rj][ij] * inv;
                if (q == qj && c <= cj) u[q] = 0;       // row j of U, strictly right of the diagonal
                a4[q][Rj][ij] = u[q];
                nu[q] = -u[q];
            }
            if constexpr (!RHS_IN) {
                const float zj = row_bcast<cj>(bz[qj]) * inv;
#pragma unroll
                for (int q = qj; q < NT; q++) bz[q] = fmaf(-u[q], zj, bz[q]);
                if (c == cj) bz[qj] = zj;
            }
            if (c == cj) dinv[qj] = inv;
            static_for<NR>([&](auto Rx) {
                constexpr int R = decltype(Rx)::value, qs = (4 * R) >> 4;
                if constexpr (4 * R + 3 > j) {
                    const float nus = row_quad_bcast<R & 3>(nu[qs]);     // -u[4R + (lane & 3)]
#pragma unroll
                    for (int q = qs; q < NT; q++) a4[q][R] = __builtin_amdgcn_mfma_f32_4x4x1f32(nus, u[q], a4[q][R], 0, 0, 0);
                }
            });
        }
    });
    // back substitution U x = z, row-oriented, reduction inside the 16-lane row (DPP)
#pragma unroll
    for (int q = 0; q < NT; q++) x[q] = 0;
    if constexpr (RHS_IN) {
        if (c == 15) x[NT - 1] = -1.0f;                 // the rhs column as an unknown fixed at -1
#pragma unroll
        for (int q = 0; q < NT; q++) bz[q] = 0;
    }
    if constexpr (!(ABL & 4))
    static_for<KMAX>([&](auto Jr) {
        constexpr int j = KMAX - 1 - decltype(Jr)::value, qj = j >> 4, cj = j & 15, Rj = j >> 2, ij = j & 3;
        {
            float part = 0;
#pragma unroll
            for (int q = qj; q < NT; q++) part = fmaf(a4[q][Rj][ij], x[q], part);
            const float sum = row16_allsum_dpp(part);
            const float xv = (bz[qj] - sum) * dinv[qj];
            if (c == cj) x[qj] = xv;
        }
    });
}

// Stream over the (up to) four rows of a quad: each row padded to a multiple of D groups.
struct QuadStream {
    uint32_t pr[5];   // CSR pointers of the quad's rows (wave-uniform)
    uint32_t step;    // entries per iteration = 4 * D
    uint32_t long_m1; // (split threshold - 1): rows of at least that many entries belong to the split path (split_rows below)
    __device__ __forceinline__ uint32_t row_ptr_at(int i) const {
        return i == 0 ? pr[0] : i == 1 ? pr[1] : i == 2 ? pr[2] : i == 3 ? pr[3] : pr[4];
    }
    // rows this stream leaves alone: empty ones (trmf.cpp:374) and long ones -- one unsigned compare: len - 1 wraps for len = 0
    __device__ __forceinline__ bool skips(int r) const { return row_ptr_at(r + 1) - row_ptr_at(r) - 1u >= long_m1; }
    __device__ __forceinline__ GramDesc first() const {
        GramDesc d{0, 0, -1};
        for (int r = 0; r < 4; r++)
            if (!skips(r)) { d = GramDesc{row_ptr_at(r), row_ptr_at(r + 1), r}; break; }
        return d;
    }
    __device__ __forceinline__ void operator()(GramDesc &d) const {
        if (d.row < 0) return;
        d.e0 += step;
        if (d.e0 < d.end) return;
        int r = d.row + 1;
        while (r < 4 && skips(r)) r++;
        if (r < 4) d = GramDesc{row_ptr_at(r), row_ptr_at(r + 1), r};
        else d = GramDesc{0, 0, -1};
    }
};

// ABL: compile-time ablation mask for profiling (bit0 skip Gram, bit1 skip factorisation, bit2 skip back solve)
// 3 wavefronts per SIMD (<= 168 VGPRs): measured 551 -> 481 us at config 3 versus the unconstrained 192
#ifndef TRMF_QUAD_WAVES
#define TRMF_QUAD_WAVES 3
#endif
// Four column tiles (rank 49..64) need more than the 168 registers of three wavefronts per SIMD: at that bound the <4,56> and
// <4,64> instantiations spilled 296 / 460 bytes per lane; they are built for two wavefronts per SIMD instead.
// (profiles/r04_fsolve_k64_fp32.txt has the comparison).
constexpr int quad_waves(int NT) { return NT >= 4 ? 2 : TRMF_QUAD_WAVES; }
#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int NT, int KMAX, int ABL = 0>
__global__ void fsolve_quad_kernel(const uint32_t *__restrict__ ptr,
                                                          const uint32_t *__restrict__ idx,
                                                          const float *__restrict__ val,
                                                          const float *__restrict__ X,
                                                          float *__restrict__ F, uint32_t row_begin,
                                                          uint32_t row_end, int k, float lambda,
                                                          uint32_t zero_row, uint32_t long_m1);
template <int NT, int KMAX>
__global__ void fsolve_quad_long_kernel(SplitRows sp, float *__restrict__ F, int k, float lambda);
#else
// LONG: four SPLIT rows per wavefront (positions 4 q .. 4 q + 3 of the long-row list): each system is the sum of its row's partials
// (split_reduce_kernel -> load_row_sum) instead of a stream of entries; finalisation, factorisation and substitutions are the same code.
template <int NT, int KMAX, int ABL, bool LONG>
__device__ __forceinline__ void fsolve_quad_body(const uint32_t *__restrict__ ptr,
                                                          const uint32_t *__restrict__ idx,
                                                          const float *__restrict__ val,
                                                          const float *__restrict__ X,
                                                          float *__restrict__ F, uint32_t row_begin,
                                                          uint32_t row_end, int k, float lambda,
                                                          uint32_t zero_row, uint32_t long_m1, const SplitRows &sp) {
    static_assert(sizeof(real) == 4, "quad F-solve is the fp32 path");
    constexpr int KP = kTile * NT;
    constexpr int SLAB = quad_slab_floats<NT>();
    __shared__ __attribute__((aligned(16))) float lds_slab[4][SLAB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 4, c = lane & 15;
    // first row of the quad (LONG: first position of the long-row list)
    const uint32_t row0 = LONG ? sp.begin + (blockIdx.x * 4u + (uint32_t)wave) * 4u : row_begin + (blockIdx.x * 4u + (uint32_t)wave) * 4u;
    if (row0 >= (LONG ? sp.end : row_end)) return;      // wave-uniform; no block barrier below
    float *S = lds_slab[wave];
    typedef float f4 __attribute__((ext_vector_type(4)));

    // a4[q][R][i] = A[4R+i][16q + c] of this lane row's system; only rows <= 16q+15 are ever touched
    quad_f4 a4[NT][KMAX / 4];
    float bz[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) {
        bz[q] = 0;
#pragma unroll
        for (int s = 0; s < KMAX; s++) a4[q][s >> 2][s & 3] = (s == kTile * q + c) ? 1.0f : 0.0f;   // idle rows: identity
    }
    bool mine = false;

    QuadStream stream;
    stream.step = 4u * kRingDepth;
    stream.long_m1 = long_m1;
    {   // row pointers of the quad (LONG: the item ranges of its four split rows) as wave-uniform scalars
        const uint32_t rr = row0 + (uint32_t)(lane < 5 ? lane : 4);
        const uint32_t lim = LONG ? sp.end : row_end;
        const uint32_t v = (LONG ? sp.first : ptr)[rr < lim ? rr : lim];
#pragma unroll
        for (int i = 0; i < 5; i++) stream.pr[i] = (uint32_t)__builtin_amdgcn_readlane((int)v, i);
    }

    GramState<NT> st;
    st.clear();
    // system `sys` is complete in st: + lambda on the diagonal (trmf.cpp:393), accumulators -> slab
    // (4 consecutive rows per store), then lane row `sys` pulls its columns into registers
    // rhs in the pad columns of the panel (quad_inject) when the rank leaves >= 8 pad columns: b_s = A[s][KP-1],
    // which the factorisation then carries along as an ordinary column (quad_factor_solve, RHS_IN)
    constexpr bool PAD = KMAX <= kTile * NT - 8;
    auto finalize = [&](int sys) {
        if constexpr (!PAD) {
#pragma unroll
            for (int q = 0; q < NT; q++) {
                st.b[q] += __shfl_xor(st.b[q], 16, kWave);
                st.b[q] += __shfl_xor(st.b[q], 32, kWave);
            }
        }
        int t = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int tj = ti; tj < NT; tj++, t++) {
                f4 v = st.acc[t];
                if (ti == tj) {
#pragma unroll
                    for (int r = 0; r < 4; r++)   // pad rows get a unit diagonal: their steps are no-ops
                        if (c == 4 * grp + r) v[r] += (kTile * ti + c < k) ? lambda : 1.0f;
                }
                *reinterpret_cast<f4 *>(&S[quad_slab_col_offset<NT>(tj) + c * (kTile * (tj + 1) + 4) + kTile * ti + 4 * grp]) = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (grp == sys) {
            mine = true;
#pragma unroll
            for (int q = 0; q < NT; q++) {
                if constexpr (!PAD) bz[q] = st.b[q];
#pragma unroll
                for (int s4 = 0; s4 < KMAX / 4; s4++) {
                    if (4 * s4 <= kTile * q + 15)
                        a4[q][s4] = *reinterpret_cast<const f4 *>(&S[quad_slab_col_offset<NT>(q) + c * (kTile * (q + 1) + 4) + 4 * s4]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        st.clear();
    };

    if constexpr (LONG) {
        for (int sys = 0; sys < 4; sys++) {             // wave-uniform: item ranges of positions past the list's end are empty
            if (stream.row_ptr_at(sys + 1) == stream.row_ptr_at(sys)) continue;
            load_row_sum<NT, !PAD>(st, sp, stream.row_ptr_at(sys), lane);
            finalize(sys);
        }
    } else if constexpr (!(ABL & 1)) {
        float nowq[NT];
#pragma unroll
        for (int q = 0; q < NT; q++) nowq[q] = 0;
        gram_ring<NT, kRingDepth, true, false, PAD>(st, idx, val, X, zero_row, 4u, lane, nowq, stream.first(), stream,
                                                    finalize);
    }

    float x[NT];
    quad_factor_solve<NT, KMAX, PAD, ABL>(a4, bz, x, c);
    if (mine) {     // logical columns {c, 16+c, ...} are adjacent in the interleaved layout: one vector store
        RealVec<NT> o;
#pragma unroll
        for (int q = 0; q < NT; q++) o.v[q] = (kTile * q + c < k) ? x[q] : 0.0f;
        const uint32_t orow = LONG ? sp.rows[row0 + (uint32_t)grp] : row0 + (uint32_t)grp;
        *reinterpret_cast<RealVec<NT> *>(F + (size_t)orow * KP + NT * c) = o;
    }
}
template <int NT, int KMAX, int ABL = 0>
__global__ __launch_bounds__(256, quad_waves(NT)) void fsolve_quad_kernel(const uint32_t *__restrict__ ptr,
                                                          const uint32_t *__restrict__ idx,
                                                          const float *__restrict__ val,
                                                          const float *__restrict__ X,
                                                          float *__restrict__ F, uint32_t row_begin,
                                                          uint32_t row_end, int k, float lambda,
                                                          uint32_t zero_row, uint32_t long_m1) {
    fsolve_quad_body<NT, KMAX, ABL, false>(ptr, idx, val, X, F, row_begin, row_end, k, lambda, zero_row, long_m1, SplitRows{});
}
// (one wavefront per SIMD less than the row kernel: the partial sums want registers beside the four systems, and a launch of this
// kernel is a few hundred wavefronts at most)
template <int NT, int KMAX>
__global__ __launch_bounds__(256, quad_waves(NT) - 1) void fsolve_quad_long_kernel(SplitRows sp, float *__restrict__ F, int k, float lambda) {
    fsolve_quad_body<NT, KMAX, 0, true>(nullptr, nullptr, nullptr, nullptr, F, 0u, 0u, k, lambda, 0u, 0u, sp);
}
#endif

#endif  // TRMF_F32

// ---- X-side Gram cache: one wavefront per timestamp row --------------------------------------------
// G_i = sum_{j in Omega_i} h_j h_j^T (k x k, full symmetric, row-major) and b_i = sum_j y_ij h_j, straight
// from the accumulator registers: diagonal tiles hold both triangles, an off-diagonal tile is written
// twice (as is and mirrored).  PACKED (the unfused X-solve, whose product is the only reader): the upper triangle
// only, row s at s k - s (s - 1) / 2, `gs` elements per timestamp.  No LDS, no workgroup barrier: a timestamp has ~nnz/T entries (1000 at
// config 3), so one wavefront amortises the ring's two-iteration lead 60x instead of 15x, and the four
// wavefronts of a workgroup never wait for each other.
#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int NT, bool RHS_PAD, bool PACKED>
__global__ void gram_x_kernel(const uint32_t *__restrict__ ptr,
                                                     const uint32_t *__restrict__ idx,
                                                     const real *__restrict__ val,
                                                     const real *__restrict__ Hf,
                                                     real *__restrict__ G, real *__restrict__ Bv,
                                                     uint32_t row_begin, uint32_t row_end, int k,
                                                     uint32_t zero_row, size_t gs, uint32_t long_thresh, const uint32_t *__restrict__ order);
template <int NT, bool RHS_PAD, bool PACKED>
__global__ void gram_x_long_kernel(SplitRows sp, real *__restrict__ G, real *__restrict__ Bv, int k, size_t gs);
template <int NT, bool RHS_PAD>
__global__ void gram_part_kernel(const uint32_t *__restrict__ idx, const real *__restrict__ val, const real *__restrict__ X,
                                 const uint32_t *__restrict__ items, uint32_t item_begin, uint32_t item_end,
                                 real *__restrict__ slab, uint32_t stride, uint32_t zero_row);
#else
// One wavefront per ITEM of a split row (entries [items[2 i], items[2 i + 1]) of one row): the partial Gram (+ rhs) of those
// entries, stored in the accumulator layout -- tile t of item i at slab[i * stride + 256 t + 4 lane ..], the rhs partials of the
// lane groups (!RHS_PAD) behind the tiles.  X / zero_row: the gathered factor and its all-zero pad row (F side: W, T; X side: H, n).
template <int NT, bool RHS_PAD>
__global__ __launch_bounds__(256) void gram_part_kernel(const uint32_t *__restrict__ idx, const real *__restrict__ val,
                                                        const real *__restrict__ X, const uint32_t *__restrict__ items,
                                                        uint32_t item_begin, uint32_t item_end, real *__restrict__ slab,
                                                        uint32_t stride, uint32_t zero_row) {
    typedef typename Mfma16<real>::acc_t acc_t;
    constexpr int NTT = NT * (NT + 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t item = item_begin + blockIdx.x * 4u + (uint32_t)wave;
    if (item >= item_end) return;                       // wave-uniform; no block barrier below
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)items[2 * item]);
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)items[2 * item + 1]);
    GramState<NT> st;
    st.clear();
    real nowq[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) nowq[q] = 0;
    if (p1 > p0)
        gram_ring<NT, kRingDepth, true, false, RHS_PAD>(st, idx, val, X, zero_row, 4u, lane, nowq, GramDesc{p0, p1, 0},
                                                        SingleRowStream{4u * kRingDepth}, [](int) {});
    real *p = slab + (size_t)item * stride;
#pragma unroll
    for (int t = 0; t < NTT; t++) *reinterpret_cast<acc_t *>(p + t * 4 * kWave + 4 * lane) = st.acc[t];
    if constexpr (!RHS_PAD) {
#pragma unroll
        for (int q = 0; q < NT; q++) p[NTT * 4 * kWave + q * kWave + lane] = st.b[q];
    }
}

// LONG: G_i / b_i of a split timestamp row from the sum of its partials (one wavefront per long row)
template <int NT, bool RHS_PAD, bool PACKED, bool LONG>
__device__ __forceinline__ void gram_x_body(const uint32_t *__restrict__ ptr,
                                                     const uint32_t *__restrict__ idx,
                                                     const real *__restrict__ val,
                                                     const real *__restrict__ Hf,
                                                     real *__restrict__ G, real *__restrict__ Bv,
                                                     uint32_t row_begin, uint32_t row_end, int k,
                                                     uint32_t zero_row, size_t gs, uint32_t long_thresh, const SplitRows &sp,
                                                     const uint32_t *__restrict__ order) {
    constexpr int KP = kTile * NT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    GramState<NT> st;
    uint32_t row;
    if constexpr (LONG) {
        const uint32_t li = sp.begin + blockIdx.x * 4u + (uint32_t)wave;
        if (li >= sp.end) return;                       // wave-uniform; no block barrier below
        row = (uint32_t)__builtin_amdgcn_readfirstlane((int)sp.rows[li]);
        load_row_sum<NT, !RHS_PAD>(st, sp, (uint32_t)__builtin_amdgcn_readfirstlane((int)sp.first[li]), lane);
    } else {
    // one wavefront per timestamp; a workgroup is 4 wavefronts, or ONE where the rows' lengths differ widely (session_xphase.hpp:
    // a workgroup's wavefront slots are released together, so three short rows would wait for the fourth)
    // `order` (skewed row lengths only): the rows longest first -- the hardware hands workgroups to SIMDs as slots free up, so with the long
    // rows dispatched first the SIMDs' sums even out (longest-processing-time-first); in index order the busiest SIMD of a power-law pattern
    // carries ~1.5x the mean (profiles/r06_split_rows.txt).  Each row's arithmetic is unchanged.
    row = row_begin + blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave;
    if (row >= row_end) return;                         // wave-uniform; no block barrier below
    if (order) row = (uint32_t)__builtin_amdgcn_readfirstlane((int)order[row]);
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ptr[row]);
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ptr[row + 1]);
    if (p1 - p0 >= long_thresh) return;                 // a split row: gram_x_long_kernel writes its G_i / b_i

    st.clear();
    real nowq[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) nowq[q] = 0;
    // no per-entry residual here: the loss at w comes out of the gradient kernel as
    // sum(y^2) + sum_i (w_i^T G_i w_i - 2 b_i.w_i)  (HV_CG_FIRST / cg_init_kernel)
    if (p1 > p0)
        gram_ring<NT, kRingDepth, true, false, RHS_PAD>(st, idx, val, Hf, zero_row, 4u, lane, nowq, GramDesc{p0, p1, 0},
                                                        SingleRowStream{4u * kRingDepth}, [](int) {});
    }
    if constexpr (RHS_PAD) {                            // rhs = column KP-1 of the panel Gram: tiles (ti, NT-1), lane column 15
        if (c == 15) {
#pragma unroll
            for (int ti = 0; ti < NT; ti++) {
                const int t = ti * NT - ti * (ti - 1) / 2 + (NT - 1 - ti);         // index of tile (ti, NT-1)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int tt = kTile * ti + Mfma16<real>::row(lane, r);
                    Bv[(size_t)row * KP + tt] = (tt < k) ? st.acc[t][r] : real(0);
                }
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < NT; q++) {                  // rhs: fold the 4 lane groups (logical column 16q + c)
            real v = st.b[q];
            v += __shfl_xor(v, 16, kWave);
            v += __shfl_xor(v, 32, kWave);
            if (g == 0) Bv[(size_t)row * KP + kTile * q + c] = (kTile * q + c < k) ? v : real(0);
        }
    }
    real *Grow = G + (size_t)row * gs;
    int t = 0;
#pragma unroll
    for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int tj = ti; tj < NT; tj++, t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rw = kTile * ti + Mfma16<real>::row(lane, r), cl = kTile * tj + c;
                if (rw < k && cl < k) {
                    const real a = st.acc[t][r];
                    if constexpr (PACKED) {
                        if (rw <= cl) Grow[rw * k - rw * (rw - 1) / 2 + cl - rw] = a;
                    } else {
                        Grow[rw * k + cl] = a;
                        if (ti != tj) Grow[cl * k + rw] = a;
                    }
                }
            }
}
template <int NT, bool RHS_PAD, bool PACKED>
__global__ __launch_bounds__(256) void gram_x_kernel(const uint32_t *__restrict__ ptr,
                                                     const uint32_t *__restrict__ idx,
                                                     const real *__restrict__ val,
                                                     const real *__restrict__ Hf,
                                                     real *__restrict__ G, real *__restrict__ Bv,
                                                     uint32_t row_begin, uint32_t row_end, int k,
                                                     uint32_t zero_row, size_t gs, uint32_t long_thresh, const uint32_t *__restrict__ order) {
    gram_x_body<NT, RHS_PAD, PACKED, false>(ptr, idx, val, Hf, G, Bv, row_begin, row_end, k, zero_row, gs, long_thresh, SplitRows{}, order);
}
template <int NT, bool RHS_PAD, bool PACKED>
__global__ __launch_bounds__(256) void gram_x_long_kernel(SplitRows sp, real *__restrict__ G, real *__restrict__ Bv, int k, size_t gs) {
    gram_x_body<NT, RHS_PAD, PACKED, true>(nullptr, nullptr, nullptr, nullptr, G, Bv, 0u, 0u, k, 0u, gs, 0u, sp, nullptr);
}
#endif

// ---- loss only (f(w_new) of the TRON acceptance test, rf_tron.h:191) ------------------------------
#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int NT>
__global__ void loss_kernel(const uint32_t *__restrict__ ptr,
                                                   const uint32_t *__restrict__ idx,
                                                   const real *__restrict__ val,
                                                   const real *__restrict__ Hf,
                                                   const real *__restrict__ W,
                                                   double *__restrict__ lossrow,
                                                   uint32_t row_begin, uint32_t row_end,
                                                   uint32_t zero_row);
#else
template <int NT>
__global__ __launch_bounds__(256) void loss_kernel(const uint32_t *__restrict__ ptr,
                                                   const uint32_t *__restrict__ idx,
                                                   const real *__restrict__ val,
                                                   const real *__restrict__ Hf,
                                                   const real *__restrict__ W,
                                                   double *__restrict__ lossrow,
                                                   uint32_t row_begin, uint32_t row_end,
                                                   uint32_t zero_row) {
    constexpr int KP = kTile * NT;
    __shared__ double Sl[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15;
    const uint32_t row = row_begin + blockIdx.x;
    if (row >= row_end) return;
    const uint32_t p0 = ptr[row], p1 = ptr[row + 1];
    GramState<NT> st;
    st.clear();
    real wq[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) wq[q] = W[(size_t)row * KP + NT * c + q];
    {
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p0 + 4u * (uint32_t)wave));
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p1);
        GramDesc d0{e0, e1, e0 < e1 ? 0 : -1};
        gram_ring<NT, kRingDepth, false, true>(st, idx, val, Hf, zero_row, 16u, lane, wq, d0,
                                               SingleRowStream{16u * kRingDepth}, [](int) {});
    }
    double l = (c == 0) ? st.loss : 0.0;
    l = wave_allsum(l);
    if (lane == 0) Sl[wave] = l;
    __syncthreads();
    if (threadIdx.x == 0) lossrow[row] = (Sl[0] + Sl[1]) + (Sl[2] + Sl[3]);
}
#endif

}  // namespace trmf
