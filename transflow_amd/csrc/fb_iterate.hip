// A3 + A4 (+ A5) of the Farneback path as ONE kernel per iteration: the 2x2 systems never leave the CU (DESIGN.md
// section 3).  Producer waves make rows of M from the expansions and the flow and keep OpenCV's column sums over an LDS
// ring; consumer waves add them across the window, solve and store the flow.  With the pre-pass that gives row segments
// their column sums' carries, and the planner that cuts a march into segments (optflowgf.cpp FarnebackUpdateMatrices +
// FarnebackUpdateFlow_Blur; cv.py:479-490).
#include "fb_common.h"

namespace {

// ---------------------------------------------------------------------------------
// One pixel of A3 split into "issue the loads" and "finish the arithmetic" (one column per lane),
// so a marching wave can keep the gathers of later rows in flight: used by the producers of
// k_flow_iter_pc below.  Same statements as update_matrix_px, rounding for rounding (see gather1_finish).
// ---------------------------------------------------------------------------------
struct Gather1 {
    float4u r0;                  // R0 at the pixel: (c0, c1, c2, c3)
    float r0c;                   // ... and c4
    float4u t0, t1, b0, b1;      // R1 on rows y1 / y1 + 1 at x1 and x1 + 1: (c0, c1, c2, c3)
    float2u t4, b4;              // c4 at x1, x1 + 1
    float dx, dy, fx, fy;
    bool inb;
};
// The wave-uniform plane bases stay in SGPRs and every load is base + one 32-bit byte offset per lane: the quad
// planes share one offset (16 bytes per pixel), the c4 planes another (4 bytes per pixel).
struct PlaneBases {
    const float *r0_q, *r0_4;
    const float *r1_q, *r1_q1, *r1_4;      // the quads at x1 and at x1 + 1 (one pixel = 16 bytes further)
    const float *r1_qb, *r1_q1b, *r1_4b;   // the same one row down
};
__device__ __forceinline__ PlaneBases plane_bases(const float *R0, const float *R1, size_t Nk, int Wk)
{
    return PlaneBases{R0, R0 + r_off4(Nk), R1, R1 + 4, R1 + r_off4(Nk), R1 + 4 * Wk, R1 + 4 * Wk + 4, R1 + r_off4(Nk) + Wk};
}

__device__ __forceinline__ float ld_f32(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ float2u ld_f32x2(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float2u *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ float4u ld_f32x4(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float4u *>(reinterpret_cast<const char *>(base) + byte_off);
}

// The addresses and weights of one pixel's gathers (everything that depends on the flow), apart from the loads.
struct GatherPrep {
    unsigned o16, o4, q16, q4; // byte offsets: the pixel in the 16-byte / 4-byte plane of R0, the top-left tap in R1's
    float dx, dy, fx, fy;
    bool inb;
};
__device__ __forceinline__ GatherPrep gather1_prep(int Wk, int Hk, int x, int y, float2 fl)
{
    GatherPrep p;
    const unsigned o = (unsigned)y * Wk + x;
    p.o16 = o * 16u;
    p.o4 = o * 4u;
    const float fx = x + fl.x, fy = y + fl.y;
    const float flx = floorf(fx), fly = floorf(fy); // (float)(int)floor(f) == floor(f) wherever the int exists
    const int x1 = (int)flx, y1 = (int)fly;
    p.dx = fl.x;
    p.dy = fl.y;
    p.fx = fx - flx;
    p.fy = fy - fly;
    p.inb = (unsigned)x1 < (unsigned)(Wk - 1) && (unsigned)y1 < (unsigned)(Hk - 1);
    // out-of-frame taps load from a clamped (valid) address and are discarded: no branch around the loads
    // (rows and widths are far below 2^24: the 24-bit multiply-add is exact and a single full-rate instruction)
    const unsigned qt = __umul24((unsigned)med3i(y1, 0, Hk - 2), (unsigned)Wk) + (unsigned)med3i(x1, 0, Wk - 2);
    p.q16 = qt * 16u;
    p.q4 = qt * 4u;
    return p;
}
__device__ __forceinline__ void gather1_load(Gather1 &g, const PlaneBases &pb, const GatherPrep &p)
{
    g.r0 = ld_f32x4(pb.r0_q, p.o16);
    g.r0c = ld_f32(pb.r0_4, p.o4);
    g.dx = p.dx;
    g.dy = p.dy;
    g.fx = p.fx;
    g.fy = p.fy;
    g.inb = p.inb;
    g.t0 = ld_f32x4(pb.r1_q, p.q16);
    g.t1 = ld_f32x4(pb.r1_q1, p.q16);
    g.b0 = ld_f32x4(pb.r1_qb, p.q16);
    g.b1 = ld_f32x4(pb.r1_q1b, p.q16);
    g.t4 = ld_f32x2(pb.r1_4, p.q4);
    g.b4 = ld_f32x2(pb.r1_4b, p.q4);
}
__device__ __forceinline__ void gather1_issue(Gather1 &g, const PlaneBases &pb, int Wk, int Hk, int x, int y, float2 fl)
{
    gather1_load(g, pb, gather1_prep(Wk, Hk, x, y, fl));
}

// update_matrix_px's statements on the channel pairs as they were loaded: per channel the bilinear sum is
// ((a00 * t(x1) + a01 * t(x1 + 1)) + a10 * b(x1)) + a11 * b(x1 + 1), two channels per packed instruction, every
// operation rounded on its own (no contraction: the file is built with -ffp-contract=off), so this M is
// k_update_matrices' and the CPU path's bit for bit.  (Round 2 let the compiler fuse these multiply-adds, +1.8 %
// frames/s at 4K; M then differed in the last bit, which was half of why border pixels flipped: DESIGN.md section 4.)
// wx, wy: the edge weights of the column and of the row.  FarnebackUpdateMatrices multiplies border[x] (x < 5),
// border[W-1-x] (x >= W-5), border[y], border[H-1-y]; from 10 x 10 up at most one factor per direction differs
// from 1 (and a factor of exactly 1 changes nothing), so the product is border(min(x, W-1-x)) * border(min(y,
// H-1-y)), the column's factor first as in the original: bit-identical, with the column's half a constant of the
// march and the row's half wave-uniform.
// UNWEIGHTED: the pixel is at least 5 pixels from every edge of the level (weight exactly 1: the five multiplications by
// it change no bit and are not issued).
template <bool UNWEIGHTED = false>
__device__ __forceinline__ void gather1_finish(const Gather1 &g, float wx, float wy, float m[5])
{
    const float fx = g.fx, fy = g.fy, dx = g.dx, dy = g.dy;
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    const float2u r0a = g.r0.xy, r0b = g.r0.zw;
    float2u r23 = a00 * g.t0.xy + a01 * g.t1.xy + a10 * g.b0.xy + a11 * g.b1.xy;
    float2u r45 = a00 * g.t0.zw + a01 * g.t1.zw + a10 * g.b0.zw + a11 * g.b1.zw;
    float r6 = a00 * g.t4.x + a01 * g.t4.y + a10 * g.b4.x + a11 * g.b4.y;
    r45 = (r0b + r45) * 0.5f;
    r6 = (g.r0c + r6) * 0.25f;
    const float o6 = g.r0c * 0.5f;
    r23 = g.inb ? r23 : float2u{0.f, 0.f};
    r45 = g.inb ? r45 : r0b;
    r6 = g.inb ? r6 : o6;
    r23 = (r0a - r23) * 0.5f;
    float r2 = r23.x, r3 = r23.y, r4 = r45.x, r5 = r45.y;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if (!UNWEIGHTED) {
        const float scale = wx * wy;
        r2 *= scale;
        r3 *= scale;
        r4 *= scale;
        r5 *= scale;
        r6 *= scale;
    }
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

// ---------------------------------------------------------------------------------
// The vertical window sums, kept as FarnebackUpdateFlow_Blur keeps them.  OpenCV holds ONE running sum per column and
// channel for the whole image: a double primed with (m + 2) copies of row 0 -- a FLOAT product -- plus rows 1 .. m - 1,
// which then receives, row after row from row 0, the FLOAT difference of the row that enters the window and the row that
// leaves it (optflowgf.cpp, FarnebackUpdateFlow_Blur: `vsum[x] += srow1[x] - srow0[x]`).  Every sum so carries the roundings of all the float
// differences above it, ~1e-7 relative: a march that starts its sums afresh at a segment's first row gets other
// roundings, and that much decides FarnebackUpdateMatrices' in-frame test for the rare border pixel whose sample point
// sits within float resolution of the last row or column.  The marching kernels therefore run
// OpenCV's chain: the segment at the top of a column primes it as OpenCV does, every other segment starts from the
// chain's value after the row above it (its "carry").  Two ways to have that value:
//   mode 1 (hand-off inside the launch): the segments of a column run one after the other; a workgroup draws a ticket,
//          tickets are dealt segment-major, and a segment waits for the carry its predecessor -- an earlier ticket, so
//          resident or finished -- publishes when it is done.  The chain is then OpenCV's operation for operation: the
//          vertical sums are bit-identical to the CPU path's.  Costs nothing where a launch has more columns of
//          workgroups than the chip has slots (the predecessor is done before the successor is dispatched).
//   mode 0 (carries from a pre-pass): where a launch is small, segments must run side by side.  A first launch marches
//          every segment for its sum of differences alone (k_flow_carry_pc; k_blur_carry where M is in memory), a scan
//          adds them up along each column (k_carry_scan), and the segments read their carry.  Adding a segment's
//          differences up before adding them to the chain re-associates double additions: ~1e-16 relative.
// The host picks per launch (choose_march).
// ---------------------------------------------------------------------------------
struct ColumnCarry {
    int mode;             // 0: carry[((seg * pairs + pair) * 5 + c) * Wk + x], written by an earlier launch
                          // 1: carry[(((seg * pairs + pair) * strips + strip) * 5 + c) * 128 + column of the strip], handed over in the launch
    int segs, pairs, strips;
    double *carry;
    unsigned *flags;      // mode 1: [((seg * pairs + pair) * strips + strip) * 2 + producer wave] == epoch once that wave's carries are stored
    unsigned epoch;       // never 0; a handle counts its chained launches
    unsigned *ticket;     // mode 1: eight counters (one list of work per XCD), zeroed before the launch
    unsigned *fault;      // host-visible word (pinned): set if a wait for a carry gave up
};

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define TF_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// One wave waits for the word its predecessor stores (cdna_hip_programming.md, guideline 16: the word is written by an
// agent-scope atomic store after the storing wave drained its payload stores; polled relaxed; the payload is then read
// with agent-scope loads, which pass the L1).  Bounded: after ~5 s the wave sets the handle's fault word and goes on
// (with a wrong carry -- the host turns the fault into an error), so every wave of the grid reaches its end.
__device__ __forceinline__ void wait_for_epoch(unsigned *flag, unsigned epoch, unsigned *fault)
{
    gu32 *f = (gu32 *)flag;
    if (__hip_atomic_load(f, TF_RLX_AGENT) != epoch) {
        const unsigned long long t0 = wall_clock64(); // 100 MHz
        while (__hip_atomic_load(f, TF_RLX_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > 500000000ull) {
                __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (no instruction: the loads that follow stay behind the poll)
}

// ---------------------------------------------------------------------------------
// One column per lane, row after row of the 2x2 systems M (A3, with A5 on the fly): what the producer waves of
// k_flow_iter_pc and the lanes of k_flow_carry_pc do.  The gathers of row e + 1 are in flight while row e is finished
// (two or three rows ahead were measured slower, on a full chip -- 2.94 -> 3.37 ms per level-0 launch at 4K x 32 -- and on a
// part-empty one alike -- 0.92 -> 1.08 -> 2.5 ms at level 1: a lone wave issues an instruction every ~8 cycles, and that, not
// memory latency, is what a step of ~130 instructions waits for), the flow of row e + 2 is the first load of a step, and for A5 the two
// lerps of a row's flow run one step after its four coarse loads.
// FLOW: where the iteration's input flow comes from -- 0: zero (coarsest scale), 1: flow_in, 2: A5 on the fly,
// resize(coarser flow, INTER_LINEAR) * 1/pyr_scale through `fi` (the statements of k_flow_upsample; the column's
// table entries are loaded once per lane, the row's are the same address for all lanes).
// ---------------------------------------------------------------------------------
template <int FLOW>
struct RowProducer {
    struct FlowRaw {
        float2 a, b, d, e; // FLOW == 2: the coarse flow at (sx, sy0), (sx + 1, sy0), (sx, sy1), (sx + 1, sy1); else a = the flow
        float fy;
    };
    int Wk, Hk, x;
    PlaneBases pb;
    const float2 *fin, *coarse;
    int Wc, Hc;
    const int *yofs;
    const float *yfrac;
    float mul;
    int up_sx, up_sx1;
    float up_fx;
    bool up_edge;
    float wx;          // the column's edge weight
    Gather1 G;         // the row being gathered
    FlowRaw F;         // the flow of the row after it
    int y_fin, y_iss;  // the (clamped) rows of G and F
    int sy_q;          // FLOW == 2: the table entries of the row whose flow is loaded next (scalar loads, fetched a step early)
    float fy_q;

    // rows_ofs / rows_frac: fi.yofs / fi.yfrac as __restrict__ kernel arguments of their own -- a march reads a row's
    // entries with SCALAR loads (the same address for all lanes), and the compiler only issues those for memory it can
    // prove nothing in the kernel writes; behind the ticket's atomic a pointer out of the by-value struct no longer
    // qualifies and the entries came as two vector loads per step (+6 % vector-memory instructions, +15 % time of an
    // A5 launch)
    __device__ __forceinline__ void init(const float *R, const float2 *flow_in, const FlowInit &fi, const int *rows_ofs,
                                         const float *rows_frac, int pair, size_t Nk, int Wk_, int Hk_, int x_)
    {
        Wk = Wk_;
        Hk = Hk_;
        x = x_;
        const int2 im = pair_images(fi, pair);
        pb = plane_bases(R + (size_t)im.x * 5 * Nk, R + (size_t)im.y * 5 * Nk, Nk, Wk);
        fin = FLOW == 1 ? flow_in + (size_t)pair * Nk : nullptr;
        coarse = FLOW == 2 ? fi.src + (size_t)pair * fi.Wc * fi.Hc : nullptr;
        Wc = fi.Wc;
        Hc = fi.Hc;
        yofs = rows_ofs;
        yfrac = rows_frac;
        mul = fi.mul;
        up_sx = up_sx1 = 0;
        up_fx = 0.f;
        up_edge = false;
        if (FLOW == 2) {
            up_sx = fi.xofs[x];
            up_fx = fi.xfrac[x];
            up_edge = up_sx >= Wc - 1; // resize.cpp: dx >= xmax copies S[sx]
            up_sx1 = min(up_sx + 1, Wc - 1);
        }
        wx = border_weight(min(x, Wk - 1 - x));
        sy_q = 0;
        fy_q = 0.f;
    }
    __device__ __forceinline__ void fetch_row_entries(int row)
    {
        if (FLOW == 2) {
            sy_q = yofs[row];
            fy_q = yfrac[row];
        }
    }
    __device__ __forceinline__ FlowRaw load_flow(int row) const // row: clamped to the level; FLOW == 2: its table entries in sy_q, fy_q
    {
        FlowRaw r;
        r.a = r.b = r.d = r.e = make_float2(0.f, 0.f);
        r.fy = 0.f;
        if (FLOW == 1) {
            const unsigned off = ((unsigned)row * Wk + x) * 8u;
            r.a = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(fin) + off);
        } else if (FLOW == 2) {
            const int sy = sy_q;
            r.fy = fy_q;
            const int sy0 = clampi(sy, 0, Hc - 1), sy1 = clampi(sy + 1, 0, Hc - 1);
            // the two coarse columns of a row as ONE 16-byte load from column min(sx, Wc - 2) (round 5: four 8-byte loads
            // before; 62.47 -> 61.9 ms per pass of 128 4K pairs): a lane on the last coarse column (up_edge) finds its
            // column in the upper half and never uses the lower
            const float4u q0 = *reinterpret_cast<const float4u *>(coarse + sy0 * Wc + up_sx1 - 1);
            const float4u q1 = *reinterpret_cast<const float4u *>(coarse + sy1 * Wc + up_sx1 - 1);
            r.a = make_float2(q0.x, q0.y);
            r.b = make_float2(q0.z, q0.w);
            r.d = make_float2(q1.x, q1.y);
            r.e = make_float2(q1.z, q1.w);
        }
        return r;
    }
    __device__ __forceinline__ float2 flow_of(const FlowRaw &r) const
    {
        if (FLOW != 2)
            return r.a;
        const float a0 = 1.f - up_fx;
        float2 h0 = make_float2(r.a.x * a0 + r.b.x * up_fx, r.a.y * a0 + r.b.y * up_fx);
        float2 h1 = make_float2(r.d.x * a0 + r.e.x * up_fx, r.d.y * a0 + r.e.y * up_fx);
        if (up_edge) {
            h0 = r.b; // (the last coarse column itself: the upper half of the load from Wc - 2)
            h1 = r.e;
        }
        const float b0 = 1.f - r.fy;
        return make_float2((h0.x * b0 + h1.x * r.fy) * mul, (h0.y * b0 + h1.y * r.fy) * mul);
    }
    // before the first next(): the march starts at row e0 (rows outside the level are their nearest row: replicated border)
    __device__ __forceinline__ void start(int e0)
    {
        y_fin = clampi(e0, 0, Hk - 1);
        y_iss = clampi(e0 + 1, 0, Hk - 1);
        fetch_row_entries(y_fin);
        gather1_issue(G, pb, Wk, Hk, x, y_fin, flow_of(load_flow(y_fin)));
        fetch_row_entries(y_iss);
        F = load_flow(y_iss);
        fetch_row_entries(clampi(e0 + 2, 0, Hk - 1));
    }
    // Row e of M (the e-th call after start(e0) is for row e0 + e ...: the caller passes the unclamped row).  INTERIOR
    // (compile time): rows e .. e + 3 lie inside the level, row e at least 5 rows from its top and bottom and the column
    // at least 5 from its sides, so nothing is clamped and the pixel's edge weight is 1 (x * 1.f == x: the same bits) --
    // the scalar clamps and selects of the general step and the five multiplications by the weight are not issued.
    template <bool INTERIOR>
    __device__ __forceinline__ void next(int e, float m[5])
    {
        // the row's edge weight is wave-uniform: border_weight() as scalar selects on the floats' bits (0.14f, 0.4472f, 1.f)
        const int dyb = min(y_fin, Hk - 1 - y_fin);
        const unsigned wyb = INTERIOR ? 0x3f800000u : (dyb < 2 ? 0x3e0f5c29u : (dyb < 5 ? 0x3ee4f766u : 0x3f800000u));
        // the flow of the row after next is the first load of the step: when the step ends by moving it into
        // place the wave waits for a load a whole step old, not for one it has just issued
        const int y_flow = INTERIOR ? e + 2 : clampi(e + 2, 0, Hk - 1);
        const FlowRaw Fn = load_flow(y_flow);
        fetch_row_entries(INTERIOR ? e + 3 : clampi(e + 3, 0, Hk - 1));
        gather1_finish<INTERIOR>(G, wx, __uint_as_float(wyb), m);
        gather1_issue(G, pb, Wk, Hk, x, y_iss, flow_of(F));
        F = Fn;
        y_fin = y_iss;
        y_iss = y_flow;
    }
};

// ---------------------------------------------------------------------------------
// A3+A4 fused, roles split inside the workgroup (the default on large levels).  Four waves march a strip
// of 128 columns together: waves 0-1 are PRODUCERS (one column per lane: RowProducer makes row e of M
// from R0, R1 and the flow; the lane keeps the window's 2M+1 rows of its column in an LDS ring, runs OpenCV's
// vertical running sum over it in double and publishes that sum), waves 2-3 are CONSUMERS taking turns by row
// (two columns per lane: a consumer adds the sums across columns -- pair sums through LDS, as
// k_blur_solve_wave --, solves and writes the flow; it copies what it needs of a row's sums out of s_v before
// the step's barrier and then has two steps for the rest, so the producers set the pace).  One workgroup
// barrier per row, sums double-buffered by step parity.  M is never stored: the window costs 38 KB of LDS per
// 112 output columns, 53.8 KB per workgroup, 3 workgroups = 12 waves per CU.
// A workgroup marches rows r0 .. r1 - 1 of its strip.  Its first WIN = 2M+1 steps only fill the ring (rows
// r0 - M - 1 .. r0 + M - 1: what the window of row r0 - 1 held); then the chain continues from the carry (see
// ColumnCarry) and every step slides it down one row: vs += (double)(entering row - leaving row), the difference in float.
// ---------------------------------------------------------------------------------
// (kernel experiments, -DTF_EXPERIMENT builds only: TF_PC_NOBARRIER makes the march's barriers LDS fences -- results wrong,
// time only: what the lockstep of a workgroup's four waves costs)
#if defined(TF_EXPERIMENT) && defined(TF_PC_NOBARRIER)
#define PC_BARRIER() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#else
#define PC_BARRIER() lds_barrier()
#endif
// s_v in two planes, even and odd strip columns apart (the odd plane's index XOR 8: a producer's 8-byte stores of one
// 16-lane group then fall in both halves of the 32 banks): the consumers' single columns at each end of a window and
// their pair come as conflict-free 8-byte reads where the 16-byte lane stride of a [128] row made them two-way conflicts.
// Level-0 launch at 4K x 32 (profiles/r05_pmc_lds_bank_conflicts.txt): SQ_LDS_BANK_CONFLICT 87.1 M cycles -> 0,
// SQ_LDS_IDX_ACTIVE 688 M -> 639 M, SQ_BUSY_CU_CYCLES -1.6 %; bit-identical results.  (0: the [128] rows of rounds 1-4)
#ifndef TF_PC_SVSPLIT
#define TF_PC_SVSPLIT 1
#endif
#ifndef TF_PC_CONS
#define TF_PC_CONS 2 // consumer waves per workgroup: they take turns by row (1: 1701, 2: 1723 frames/s at 4K x 32)
#endif
#define TF_PC_THREADS (128 + 64 * TF_PC_CONS)
template <int M, int FLOW>
__global__ void __launch_bounds__(TF_PC_THREADS)
k_flow_iter_pc(const float *__restrict__ R, const float2 *__restrict__ flow_in, float2 *__restrict__ flow_out, int Wk,
               int Hk, double scale, int seg, FlowInit fi, const int *__restrict__ rows_ofs, const float *__restrict__ rows_frac,
               ColumnCarry cc)
{
    static_assert(M & 1, "the pair-sum window needs an odd half-width");
    // A strip's halo is M columns rounded up to whole lanes: 112 outputs per strip for M = 7, every strip starting on
    // a multiple of 8 columns = 64 bytes of the 8-byte planes (114 outputs on strips that start on odd columns were
    // measured 3 % slower: every 512-byte row piece a wave loads then straddles one more 128-byte line).
    constexpr int HALO = (M + 1) & ~1;
    constexpr int OUTC = 128 - 2 * HALO, WIN = 2 * M + 1;
    __shared__ float ring[WIN][5][128];   // the window's rows of M, one column per producer lane
    __shared__ double s_v[2][5][128];     // vertical window sums of the row just produced (double-buffered by step parity)
    constexpr int CONS = TF_PC_CONS;
    __shared__ double s_p[CONS][5][64]; // each consumer's pair sums
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned bx, by;
    int pair;
    if (cc.mode == 1) {
        // Tickets, one list per XCD.  The launch's pairs are dealt to the eight XCDs in contiguous runs, and the workgroups
        // an XCD receives (the hardware deals workgroup i to XCD i mod 8) draw from their XCD's list first: the strips of a
        // pair stand side by side in one L2 (their halo columns are read once), and so do consecutive pairs, which read
        // the frame they share -- R0 of one, R1 of the next -- at the same rows at about the same time.  A list is
        // segment-major (then strip, then pair): every segment of a column has a later ticket in the same list than the
        // segment above it, and whoever holds a ticket is resident, so a wait for a predecessor always ends, whatever
        // order the hardware dispatches in.  A workgroup whose own list is used up takes from the next one (lists
        // differ in length when the pairs do not divide by eight): every workgroup finds exactly one ticket.
        // (the ticket travels through a double of s_p, which the consumers first touch many barriers later: a word of
        // its own would be the 43rd LDS granule of 1280 bytes and cost the CU its third workgroup)
        unsigned *s_ticket = reinterpret_cast<unsigned *>(&s_p[0][0][0]);
        if (threadIdx.x == 0) {
            unsigned got = ~0u, list = 0;
            for (unsigned k = 0; k < 8 && got == ~0u; k++) {
                list = (blockIdx.x + k) & 7;
                const unsigned n = (list + 1) * (unsigned)cc.pairs / 8 - list * (unsigned)cc.pairs / 8;
                const unsigned len = n * (unsigned)(cc.segs * cc.strips);
                if (len == 0 || __hip_atomic_load(cc.ticket + list, TF_RLX_AGENT) >= len)
                    continue; // (a look first: a list that is used up is not counted up again by every passer-by)
                const unsigned t = __hip_atomic_fetch_add(cc.ticket + list, 1u, TF_RLX_AGENT);
                if (t < len)
                    got = t;
            }
            s_ticket[0] = got;
            s_ticket[1] = list;
        }
        __syncthreads();
        const unsigned t = __builtin_amdgcn_readfirstlane(s_ticket[0]), list = __builtin_amdgcn_readfirstlane(s_ticket[1]);
        __syncthreads(); // every wave has read it
        if (t == ~0u)
            return; // (no ticket left: the grid and the lists have parted -- touch nothing)
        const unsigned p0 = list * (unsigned)cc.pairs / 8, n = (list + 1) * (unsigned)cc.pairs / 8 - p0;
        const unsigned per_seg = n * (unsigned)cc.strips;
        by = t / per_seg;
        const unsigned rem = t - by * per_seg;
        bx = rem / n;
        pair = (int)(p0 + (rem - bx * n));
    } else {
        xcd_pair_tile(bx, by, pair);
    }
    const size_t Nk = (size_t)Wk * Hk;
    const int r0 = by * seg, r1 = min(r0 + seg, Hk);
    // step s: the producers make row e = r0 - M - 1 + s of M (s < n_rows) and, from s = WIN on, the window sums of row
    // e - M; a consumer turns the sums of step s - 1 into the flow of row r0 + (s - 1) - WIN
    const int e0 = r0 - M - 1, n_rows = (r1 - r0) + WIN, nsteps = n_rows + 1;
    if (wave < 2) {
        // the producers set a step's pace: where a SIMD holds a producer and consumers (or another kernel's waves) the
        // producer issues first (4K x 32, one batch in flight: level 0 2859 -> 2806 us, level 1 894 -> 851, 1750 -> 1784
        // frames/s; with two batches in flight nothing changes)
        __builtin_amdgcn_s_setprio(1);
        const int col = wave * 64 + lane;
        const int x = clampi((int)bx * OUTC - HALO + col, 0, Wk - 1); // replicated border columns
        RowProducer<FLOW> P;
        P.init(R, flow_in, fi, rows_ofs, rows_frac, pair, Nk, Wk, Hk, x);
        P.start(e0);
        double vs[5] = {0, 0, 0, 0, 0};
        int slot = 0;
        // The ring's first WIN rows.  The segment at the top of the level also primes the chain, statement for statement
        // as FarnebackUpdateFlow_Blur does: vsum = row 0 * (m + 2), a float product; vsum += row y for y = 1 .. m - 1.
        for (int s = 0; s < WIN; s++) {
            float m[5];
            P.template next<false>(e0 + s, m);
#pragma unroll
            for (int c = 0; c < 5; c++)
                ring[slot][c][col] = m[c];
            if (r0 == 0) {
                if (s == 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++)
                        vs[c] = (double)(m[c] * (float)(M + 2));
                } else if (s >= M + 2 && s <= 2 * M) { // rows 1 .. M - 1 (beyond the last row: the last row again)
#pragma unroll
                    for (int c = 0; c < 5; c++)
                        vs[c] += (double)m[c];
                }
            }
            slot = slot + 1 == WIN ? 0 : slot + 1;
            PC_BARRIER();
        }
        if (r0 != 0) { // the chain's value after row r0 - 1
            if (cc.mode == 1) {
                const size_t item = ((size_t)by * cc.pairs + pair) * cc.strips + bx;
                wait_for_epoch(cc.flags + item * 2 + wave, cc.epoch, cc.fault);
                gu64 *C = (gu64 *)(cc.carry + item * (5 * 128) + col);
#pragma unroll
                for (int c = 0; c < 5; c++)
                    vs[c] = __longlong_as_double((long long)__hip_atomic_load(C + c * 128, TF_RLX_AGENT));
            } else {
                const double *C = cc.carry + (((size_t)by * cc.pairs + pair) * 5) * Wk + x;
#pragma unroll
                for (int c = 0; c < 5; c++)
                    vs[c] = C[(size_t)c * Wk];
            }
        }
        // One step of the march proper.  A march runs three loops: the rows at the top of the level, the interior (see
        // RowProducer::next), the rows at the bottom.
        auto step = [&](int s, auto interior) {
            constexpr bool INTERIOR = decltype(interior)::value;
            float m[5];
            P.template next<INTERIOR>(e0 + s, m);
            // the row that leaves the window sits in the slot the new row takes (e - WIN == e mod WIN);
            // only this lane ever touches its column of the ring
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float old = ring[slot][c][col];
                ring[slot][c][col] = m[c];
                vs[c] += (double)(m[c] - old); // vsum[x] += srow1[x] - srow0[x]: a float difference accumulated in double
#if TF_PC_SVSPLIT
                s_v[s & 1][c][(col & 1) * 64 + ((col >> 1) ^ ((col & 1) << 3))] = vs[c];
#else
                s_v[s & 1][c][col] = vs[c];
#endif
            }
            slot = slot + 1 == WIN ? 0 : slot + 1;
            PC_BARRIER();
        };
        // steps whose row e = e0 + s lies in [5, Hk - 6] (then e + 3 <= Hk - 1 too)
        int s_in0 = min(max(5 - e0, WIN), n_rows), s_in1 = min(max(Hk - 5 - e0, s_in0), n_rows);
        // ... and only in strips whose 128 columns all lie at least 5 pixels inside the level (wave-uniform): there the
        // interior step also drops the edge weight's five multiplications
        const int strip0 = (int)bx * OUTC - HALO;
        if (strip0 < 5 || strip0 + 127 > Wk - 6)
            s_in0 = s_in1 = WIN;
        int s = WIN;
        for (; s < s_in0; s++)
            step(s, std::false_type{});
        for (; s < s_in1; s++)
            step(s, std::true_type{});
        for (; s < n_rows; s++)
            step(s, std::false_type{});
        if (cc.mode == 1 && (int)by + 1 < cc.segs) {
            // vs is the chain after row r1 - 1: the carry of the segment below.  Write-through stores, drained, then this
            // wave's flag (each producer wave hands over its own 64 columns: no barrier between the two).
            const size_t item = ((size_t)(by + 1) * cc.pairs + pair) * cc.strips + bx;
            gu64 *C = (gu64 *)(cc.carry + item * (5 * 128) + col);
#pragma unroll
            for (int c = 0; c < 5; c++)
                __hip_atomic_store(C + c * 128, (unsigned long long)__double_as_longlong(vs[c]), TF_RLX_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store((gu32 *)(cc.flags + item * 2 + wave), cc.epoch, TF_RLX_AGENT);
        }
        PC_BARRIER(); // step n_rows: the consumers' last row
    } else {
        // The consumers take turns: wave 2 + k serves the steps with s % TF_PC_CONS == k.  In its step a consumer first
        // takes what it needs of the row's sums out of s_v (the sum of its two columns and one single column at each end of
        // the window) -- that much must be done before the step's barrier, after which the producers overwrite the
        // buffer -- and then has until its next turn for the exchange of pair sums, the solve and the store.
        const int who = wave - 2;
        double(*sp)[64] = s_p[who];
        // A lane's two columns are strip columns 2 * lane and the next one; the halo is whole lanes
        const int c0 = (int)bx * OUTC - HALO + 2 * lane;
        constexpr int first_out = HALO / 2, last_out = (128 - HALO) / 2 - 1; // lanes whose two columns are outputs
        static_assert(last_out - first_out + 1 == OUTC / 2, "outputs are whole lanes");
        const bool is_out = lane >= first_out && lane <= last_out && c0 < Wk;
        const double eps = 1e-3 / (scale * scale);
        constexpr int kk = (M + 1) / 2;
        const int lo = max(lane - kk, 0), hi = min(lane + kk, 63);
        for (int s = 0; s < nsteps; s++) {
            const int y = r0 + (s - 1) - WIN; // the row whose window the producers completed in step s - 1
            const bool mine = (s % CONS) == who && y >= r0; // wave-uniform
            double p[5], left[5], right[5];
            if (mine) {
                const double(*sv)[128] = s_v[(s - 1) & 1];
#pragma unroll
                for (int c = 0; c < 5; c++) {
#if TF_PC_SVSPLIT
                    p[c] = sv[c][lane] + sv[c][64 + (lane ^ 8)];
                    left[c] = sv[c][64 + (lo ^ 8)];
                    right[c] = sv[c][hi];
#else
                    p[c] = sv[c][2 * lane] + sv[c][2 * lane + 1];
                    left[c] = sv[c][2 * lo + 1];
                    right[c] = sv[c][2 * hi];
#endif
                }
            }
            PC_BARRIER();
            if (mine) {
                // The M pair sums of a window through sums of three: T[l] = P[l-1] + P[l] + P[l+1] replaces P in
                // LDS (a lane keeps its own P), and the window is T[l] (M = 3), T[l-1] + T[l+1] - P[l] (M = 5) or
                // T[l-2] + P[l] + T[l+2] (M = 7): four LDS accesses and four additions per channel instead of
                // eight and six.  T of lanes 0 and 63 is not a sum of three and no output lane reads it.
                // (The same exchange as whole-wave DPP shifts was measured 4 % slower: tools/variants/.)
                static_assert(M == 3 || M == 5 || M == 7, "window sums from sums of three");
                double t[5];
#pragma unroll
                for (int c = 0; c < 5; c++)
                    sp[c][lane] = p[c];
                lds_wave_sync();
                const int lm = max(lane - 1, 0), lp = min(lane + 1, 63);
#pragma unroll
                for (int c = 0; c < 5; c++)
                    t[c] = (sp[c][lm] + p[c]) + sp[c][lp];
                if (M > 3) {
                    lds_wave_sync();
#pragma unroll
                    for (int c = 0; c < 5; c++)
                        sp[c][lane] = t[c];
                    lds_wave_sync();
                }
                if (is_out) {
                    double g0[5], g1[5];
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        double common;
                        if (M == 3)
                            common = t[c];
                        else if (M == 5)
                            common = (sp[c][lane - 1] + sp[c][lane + 1]) - p[c];
                        else
                            common = (sp[c][lane - 2] + p[c]) + sp[c][lane + 2];
                        g0[c] = left[c] + common;
                        g1[c] = common + right[c];
                    }
                    float2 *o = flow_out + (size_t)pair * Nk + (size_t)y * Wk + c0;
                    {
                        // the solve with its multiply-adds fused and one Newton step on v_rcp_f64 (a float leaves here):
                        // 10 fp64 instructions less per row, +1 % frames/s; k_blur_solve_wave keeps the separate
                        // operations and the second step
#pragma clang fp contract(fast)
                        const double idet0 = fast_recip(g0[0] * g0[2] - g0[1] * g0[1] + eps);
                        const double idet1 = fast_recip(g1[0] * g1[2] - g1[1] * g1[1] + eps);
                        const float4u f = {(float)((g0[0] * g0[4] - g0[1] * g0[3]) * idet0),
                                           (float)((g0[2] * g0[3] - g0[1] * g0[4]) * idet0),
                                           (float)((g1[0] * g1[4] - g1[1] * g1[3]) * idet1),
                                           (float)((g1[2] * g1[3] - g1[1] * g1[4]) * idet1)};
                        if (c0 + 1 < Wk) // both columns in one 16-byte store (8-byte aligned where the level's width is odd)
                            *reinterpret_cast<float4u *>(o) = f;
                        else
                            o[0] = make_float2(f.x, f.y);
                    }
                }
                lds_wave_sync();
            }
        }
    }
}
// ---------------------------------------------------------------------------------
// The pre-pass of mode 0 for the one-kernel iteration (M is never in memory there): every lane marches ONE column of one
// segment through the same rows of M the iteration will make, with the same ring, for the chain's increments alone --
// no halo columns, no window sums across columns, no barrier: the waves run free.  Segment 0 delivers the chain's value
// after its last row (primed as OpenCV primes it), the others the sum of their rows' increments from zero; k_carry_scan
// turns that into each segment's carry.  S: [segment][pair][channel][Wk].
// ---------------------------------------------------------------------------------
// STORE (option fb_exact_sums on a large launch): one segment = the whole column, and what leaves is not the segment's
// last value but the chain's value at EVERY row, V[pair][channel][y][x] -- k_exact_vsum's output without M ever being in
// memory (k_exact_hsolve takes it from there).
template <int M, int FLOW, bool STORE = false>
__global__ void __launch_bounds__(128)
k_flow_carry_pc(const float *__restrict__ R, const float2 *__restrict__ flow_in, int Wk, int Hk, int seg, FlowInit fi,
                const int *__restrict__ rows_ofs, const float *__restrict__ rows_frac, double *__restrict__ S)
{
    constexpr int WIN = 2 * M + 1;
    __shared__ float ring[WIN][5][128];
    const int col = threadIdx.x, xr = blockIdx.x * 128 + col, x = min(xr, Wk - 1);
    const int by = blockIdx.y, pair = blockIdx.z;
    const size_t Nk = (size_t)Wk * Hk;
    const int r0 = by * seg, r1 = min(r0 + seg, Hk);
    const int e0 = r0 - M - 1, n_rows = (r1 - r0) + WIN;
    RowProducer<FLOW> P;
    P.init(R, flow_in, fi, rows_ofs, rows_frac, pair, Nk, Wk, Hk, x);
    P.start(e0);
    double vs[5] = {0, 0, 0, 0, 0};
    int slot = 0;
    for (int s = 0; s < WIN; s++) {
        float m[5];
        P.template next<false>(e0 + s, m);
#pragma unroll
        for (int c = 0; c < 5; c++)
            ring[slot][c][col] = m[c];
        if (r0 == 0) {
            if (s == 0) {
#pragma unroll
                for (int c = 0; c < 5; c++)
                    vs[c] = (double)(m[c] * (float)(M + 2));
            } else if (s >= M + 2 && s <= 2 * M) {
#pragma unroll
                for (int c = 0; c < 5; c++)
                    vs[c] += (double)m[c];
            }
        }
        slot = slot + 1 == WIN ? 0 : slot + 1;
    }
    double *V = S + (size_t)pair * 5 * Nk + (size_t)r0 * Wk + x; // (STORE)
    for (int s = WIN; s < n_rows; s++) {
        float m[5];
        P.template next<false>(e0 + s, m);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const float old = ring[slot][c][col];
            ring[slot][c][col] = m[c];
            vs[c] += (double)(m[c] - old);
            if (STORE && xr < Wk)
                V[(size_t)c * Nk] = vs[c];
        }
        V += Wk;
        slot = slot + 1 == WIN ? 0 : slot + 1;
    }
    if (!STORE && xr < Wk) {
        double *o = S + (((size_t)by * gridDim.z + pair) * 5) * Wk + xr;
#pragma unroll
        for (int c = 0; c < 5; c++)
            o[(size_t)c * Wk] = vs[c];
    }
}

} // namespace

namespace tf {
namespace fb {

// ---------------------------------------------------------------------------------
// How a marching launch is cut into row segments, and where its segments get the column sums' carries from
// (ColumnCarry).  `columns` = strips x pairs workgroups stand side by side; a launch of `segs` segments has segs x columns
// workgroups of (h / segs + warm-up) steps each, `slots` of them resident at a time.
//   hand-off inside the launch (mode 1): the segments of a column run one after the other.  With at least as many
//     columns as slots that costs nothing -- by the time a segment is dispatched the one above it is done -- and the
//     segment count is the one that minimises rounds x steps.  With fewer columns the segments would only queue up
//     behind each other, so the column is marched whole (one segment, no hand-off) on a part-empty chip.
//   pre-pass (mode 0): segments side by side as before, for a second launch that makes the rows of M for their carries
//     (prepass_cost x the march's time; k_blur_carry, which reads M, is cheap) and a third that adds them up: two
//     launches of a fixed cost each (prepass_steps, in steps of the march) that a short column does not repay.
// Times are in units of one workgroup step at full residency; a step is faster on a part-empty chip (step_time).
// ---------------------------------------------------------------------------------
static double step_time(double wgs_per_cu, int slots_per_cu)
{
    // measured on MI355X for k_flow_iter_pc (3 slots per CU): a step takes 0.69 / 0.83 / 0.88 us with 1.1 / 2.25 / 3
    // workgroups per CU on average (a wave's ~130 instructions per step at one issue every ~8 cycles, not memory latency,
    // set the pace, so company costs little)
    const double full = slots_per_cu, o = std::min(std::max(wgs_per_cu, 1.0), full);
    static const double alone = tune("TF_STEP_ALONE_PCT", 78) / 100.0;
    return full <= 1 ? 1.0 : alone + (1.0 - alone) * (o - 1.0) / (full - 1.0);
}
static March choose_march(long columns, int h, int warm, long slots, int slots_per_cu, double prepass_cost, double prepass_steps, int min_rows,
                          bool has_company = false)
{
    const long forced_segs = option(OPT_FB_SEGS), forced_mode = option(OPT_FB_CHAIN);
    const long cus = std::max(1l, slots / slots_per_cu);
    auto rounds_cost = [&](long sg, double *cost) {
        const long rows = (h + sg - 1) / sg;
        const long wgs = columns * sg, rounds = (wgs + slots - 1) / slots;
        *cost = (double)rounds * (double)(rows + warm + 1) * step_time((double)std::min(wgs, slots) / cus, slots_per_cu);
        return rows;
    };
    // the best segment count for segments that run side by side
    long best_segs = 1;
    double best_cost = 1e300;
    for (long sg = 1; sg <= 64 && sg <= h; sg++) {
        double cost;
        const long rows = rounds_cost(sg, &cost);
        if (rows < min_rows && sg > 1)
            break;
        if (cost < best_cost * 0.999) {
            best_cost = cost;
            best_segs = sg;
        }
    }
    if (forced_segs > 0) {
        best_segs = std::min<long>(forced_segs, h);
        rounds_cost(best_segs, &best_cost);
    }
    // A handle with a lane (tf_fb_create_lane) has the other lane's batch for company: what a whole-column march leaves idle
    // is not lost, while a pre-pass's second making of M is work the chip does not get back -- the pre-pass must win by more
    // (measured at 4K x 8, two lanes: 1444 frames/s with whole columns, 1124 with the pre-pass the lone-launch model picks)
    static const double company = tune("TF_PC_COMPANY_PCT", 70) / 100.0;
    March m;
    double whole;
    rounds_cost(1, &whole);
    if (best_segs == 1) {
        m.mode = 0;
        m.segs = 1;
    } else if (forced_mode == 1 || (forced_mode < 0 && columns >= slots)) {
        m.mode = 1;
        m.segs = (int)best_segs;
    } else if (forced_mode == 0 || forced_segs > 0 || best_cost * (1.0 + prepass_cost) + prepass_steps < whole * (has_company ? company : 1.0)) {
        m.mode = 0;
        m.segs = (int)best_segs;
    } else {
        m.mode = 0;
        m.segs = 1;
    }
    m.seg = (h + m.segs - 1) / m.segs;
    m.segs = (h + m.seg - 1) / m.seg;
    return m;
}

// `up`: the first iteration of a level below the coarsest takes its flow from the coarser level (A5 fused)
template <int M>
static int launch_flow_iter(tf_fb *fb, int w, int h, int n_pairs, const float2 *flow_in, float2 *flow_out, int k,
                            const FlowInit *up)
{
    const double scale = 1. / ((double)fb->prm.winsize * fb->prm.winsize);
    const float *R = fb->Rk(k);
    constexpr int OUTC = 128 - 2 * ((M + 1) & ~1), WIN = 2 * M + 1;
    const unsigned strips = cdiv(w, OUTC);
    FlowInit f;
    memset(&f, 0, sizeof(f));
    if (up)
        f = *up;
    f.rmap = fb->rmap_dev;
    if (fb_exact(fb)) {
        // fb_exact_sums on a large launch: the column sums of EVERY row straight from R0, R1 and the flow (the pre-pass
        // kernel with one segment, storing as it goes: M is never in memory), then the row walker.  4K x 32, level 0: 4.3 +
        // 2.7 ms against 3.3 + 3.1 + 2.8 through update-matrices (and 9.0 for an exact form of k_flow_iter_pc whose
        // consumers handed the rows' sums from strip to strip, tried in round 4: profiles/NOTES.md)
        TF_TRY(fb_exact_room(fb, w, h, n_pairs));
        const dim3 pgrid(cdiv(w, 128), 1, n_pairs);
        double *V = fb->exact_vsum.as<double>();
        const char *name = lvl_name("fb_flow_vsum", k);
        if (up)
            TF_TRY(launch(name, k_flow_carry_pc<M, 2, true>, pgrid, dim3(128), 0, R, flow_in, w, h, h, f, f.yofs, f.yfrac, V));
        else if (flow_in)
            TF_TRY(launch(name, k_flow_carry_pc<M, 1, true>, pgrid, dim3(128), 0, R, flow_in, w, h, h, f, f.yofs, f.yfrac, V));
        else
            TF_TRY(launch(name, k_flow_carry_pc<M, 0, true>, pgrid, dim3(128), 0, R, flow_in, w, h, h, f, f.yofs, f.yfrac, V));
        return fb_exact_hsolve(fb, w, h, n_pairs, flow_out, k);
    }
    // 3 workgroups per CU are resident (768 on the chip) and all take the same time: the launch runs in
    // rounds of 768, each as long as a segment plus its 2M+1 warm-up steps and the drain step (choose_march)
    static const long slots = tune("TF_PC_SLOTS", 768);
    static const double prepass_cost = tune("TF_PC_PREPASS_PCT", 80) / 100.0; // k_flow_carry_pc: 0.76 of the march it serves (4K x 32, level 2)
    March mc = choose_march((long)strips * n_pairs, h, WIN, slots, 3, prepass_cost, 14, 2 * WIN, fb->lane_of || fb->lanes); // (steps of ~0.9 us)
    ColumnCarry cc;
    memset(&cc, 0, sizeof(cc));
    cc.mode = mc.mode;
    cc.segs = mc.segs;
    cc.pairs = n_pairs;
    cc.strips = (int)strips;
    dim3 grid(strips, mc.segs, n_pairs);
    if (mc.segs > 1 && mc.mode == 1) {
        const size_t items = (size_t)mc.segs * n_pairs * strips;
        TF_TRY(fb_carry_room(fb, items * 5 * 128, items * 2));
        if (++fb->chain_epoch == 0) { // 2^32 chained launches later: the flags start over
            TF_HIP(hipMemsetAsync(fb->chain_words.as<unsigned>() + 16, 0, fb->chain_words.bytes - 64, stream()));
            fb->chain_epoch = 1;
        }
        cc.mode = 1;
        cc.carry = fb->col_carry.as<double>();
        cc.flags = fb->chain_words.as<unsigned>() + 16;
        cc.epoch = fb->chain_epoch;
        cc.ticket = fb->chain_words.as<unsigned>();
        TF_HIP(hipMemsetAsync(cc.ticket, 0, 8 * sizeof(unsigned), stream()));
        cc.fault = fb->chain_fault;
        grid = dim3((unsigned)items);
    } else if (mc.segs > 1) {
        TF_TRY(fb_carry_room(fb, (size_t)mc.segs * n_pairs * 5 * w, 0));
        cc.carry = fb->col_carry.as<double>();
        cc.fault = fb->chain_fault;
        const dim3 pgrid(cdiv(w, 128), mc.segs, n_pairs);
        if (up)
            TF_TRY(launch(lvl_name("fb_flow_carry", k), k_flow_carry_pc<M, 2>, pgrid, dim3(128), 0, R, flow_in, w, h, mc.seg, f, f.yofs, f.yfrac, cc.carry));
        else if (flow_in)
            TF_TRY(launch(lvl_name("fb_flow_carry", k), k_flow_carry_pc<M, 1>, pgrid, dim3(128), 0, R, flow_in, w, h, mc.seg, f, f.yofs, f.yfrac, cc.carry));
        else
            TF_TRY(launch(lvl_name("fb_flow_carry", k), k_flow_carry_pc<M, 0>, pgrid, dim3(128), 0, R, flow_in, w, h, mc.seg, f, f.yofs, f.yfrac, cc.carry));
        TF_TRY(fb_carry_scan(fb, w, n_pairs, mc.segs, k));
    } else {
        cc.mode = 0;
    }
    int rc;
    const int kind = up ? 2 : (flow_in ? 1 : 0);
    const char *name = lvl_name("fb_flow_iter", k);
#define TF_PC_LAUNCH(FLOWK)                                                                                                    \
    launch(name, k_flow_iter_pc<M, FLOWK>, grid, dim3(TF_PC_THREADS), 0, R, flow_in, flow_out, w, h, scale, mc.seg, f, f.yofs,   \
           f.yfrac, cc)
    rc = kind == 2 ? TF_PC_LAUNCH(2) : (kind == 1 ? TF_PC_LAUNCH(1) : TF_PC_LAUNCH(0));
#undef TF_PC_LAUNCH
    return rc;
}

bool fb_flow_iter(tf_fb *fb, int w, int h, int n_pairs, const float2 *flow_in, float2 *flow_out, int k, int &rc,
                         const FlowInit *up)
{
    if (w < 10 || h < 10) // border_scale: below 10 x 10 the two-kernel form carries OpenCV's edge test
        return false;
    switch (fb->prm.winsize / 2) { // odd half-widths: the pair-sum window
    case 3: rc = launch_flow_iter<3>(fb, w, h, n_pairs, flow_in, flow_out, k, up); return true;
    case 5: rc = launch_flow_iter<5>(fb, w, h, n_pairs, flow_in, flow_out, k, up); return true;
    case 7: rc = launch_flow_iter<7>(fb, w, h, n_pairs, flow_in, flow_out, k, up); return true;
    default: return false;
    }
}

} // namespace fb
} // namespace tf
