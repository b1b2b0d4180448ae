// MFMA GEMM for gfx950 (MI355X): C[M,N] = (Ahi [+ Alo])[M,K] . Wt[N,K]^T (+ bias) with fused epilogues.
//
// This is the dominant kernel of the hot path.  It replaces the cuBLAS `addmm` behind upstream
// jukebox `Conv1D.forward` (4 per prior layer, reached from jukebox/main.py:108) and the
// `nn.Linear`s of HF Llama / `mm_projector` (m2t/models/llamav2.py:79,133,224-234,312).
//
// Precision scheme ("split" mode, used for the Jukebox prior which the reference runs with
// fp16=False, i.e. fp32 activations x fp16-VALUED weights): the fp32 activation a is carried as two
// fp16 planes a = hi + lo (hi = fp16(a), lo = fp16(a - hi), 22 significant bits) produced by the
// previous kernel's epilogue; the weight is exactly its fp16 storage.  The kernel runs TWO
// 16-bit MFMA passes per K-step against the SAME weight fragments and accumulates in fp32, i.e.
// fp32-class accuracy at 1/2 of the fp16 matrix rate instead of 1/16 (the f32-input MFMA rate).
// Single-pass mode (SPLIT=false) is the plain bf16/fp16 GEMM used for Llama.
//
// Structure of the LDS-staged kernels in this file (`gemm_tile`, tile variants 0 / 1 / 2 / 11 / 12 and the persistent form 20;
// the default for most shapes is 128x256x64 with 4 waves of 64x128, a single 64 KiB stage and two workgroups per CU): waves own
// TM x TN tiles of v_mfma_f32_32x32x16_{f16,bf16}; operands stream HBM/L2 -> LDS with global_load_lds_dwordx4 (16 B/lane, no
// VGPR round trip) into an LDS image whose 16-B chunks are XOR swizzled on the SOURCE address (the DMA destination is
// lane-linear) so that the ds_read_b128 fragment reads are bank-conflict free; one barrier per K-step.  blockIdx is remapped
// XCD-aware (each XCD's private L2 sees a contiguous band of tiles) and grouped over M so that co-resident blocks share A and
// W panels.  Also here: the B-direct kernels (weights fragment-major, L2 -> VGPR; the Llama prefill) with their K-splitting
// forms, and the skinny weight-streaming kernels of the decode step.  The prior's 256x256 tiles live in gemm256n.hip (two-pass
// fp16, default), gemm256.hip (its predecessor, the M-split LDS ring) and gemm256_lo8n.hip (fp8 low plane, opt-in).

// Profiling-only compile-time ablations (results invalid): 1 = no DMA in the K loop, 2 = no LDS fragment
// reads, 3 = no MFMA.  Built into separate libraries by scripts/build_ablations.sh; never set in the product.
#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0
#endif

#include <new>

#include "gemm_core.h"
#include "gemm_rope_epi.h"
#include "gemm_bd.h"
#include "gemm_bda_loop.h"

namespace llark {

// One output tile (linear tile index `bid` in the M-grouped order) computed by the calling workgroup.
template <typename T, bool SPLIT, int EPI, typename C>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, int bid, char* smem) {
    typedef typename Mfma<T>::frag frag;
    constexpr int STAGE = (SPLIT ? 2 : 1) * C::A_BYTES + C::B_BYTES;     // [Ahi, (Alo), W]
    constexpr int OFF_L = C::A_BYTES, OFF_W = (SPLIT ? 2 : 1) * C::A_BYTES;
    constexpr int NS = C::NSTAGE;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / C::WN, wn = w % C::WN;

    // M-grouped tile order: groups of GM tile rows, N-major inside a group, so that tiles with neighbouring
    // indices share A / W panels (speed only; any mapping is correct)
    constexpr int GM = 8;
    const int gsz = GM * p.tiles_n;
    const int g = bid / gsz;
    const int first_m = g * GM;
    const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
    const int tile_m = first_m + (bid % gsz) % gm;
    const int tile_n = (bid % gsz) / gm;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;

    const long long bz = blockIdx.y;                              // batch index (0 when not batched)
    const T* Ahi = (const T*)p.Ahi + bz * p.sA;
    const T* Alo = SPLIT ? (const T*)p.Alo + bz * p.sA : nullptr;
    const T* Wt = (const T*)p.Wt + bz * p.sW;

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
        const int k0 = kt * C::BK;
#pragma unroll
        for (int i = 0; i < (C::BM / C::RPI) / C::NW; ++i) {
            const int j = w + i * C::NW;                                   // DMA instruction index in the A tile
            dma_rows<T, C>(Ahi, p.lda, m0 + j * C::RPI, p.M, k0, j * C::RPI, base + j * 1024, lane);
            if (SPLIT) dma_rows<T, C>(Alo, p.lda, m0 + j * C::RPI, p.M, k0, j * C::RPI, base + OFF_L + j * 1024, lane);
        }
#pragma unroll
        for (int i = 0; i < (C::BN / C::RPI) / C::NW; ++i) {
            const int j = w + i * C::NW;
            dma_rows<T, C>(Wt, p.ldw, n0 + j * C::RPI, p.N, k0, j * C::RPI, base + OFF_W + j * 1024, lane);
        }
    };

    f32x16_t acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.Kp / C::BK;
    if (NS >= 2) stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        if (NS == 1) {
            // single LDS stage, two barriers per K-step: overlap comes from the other blocks resident on the CU
            if (kt) __syncthreads();
            stage(0, kt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else if (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk && GEMM_ABLATE != 1) stage((kt + 1) & 1, kt + 1);
        }
        const char* base = smem + (NS == 1 ? 0 : (kt & 1)) * STAGE;
        const char* sA = base;
        const char* sL = base + OFF_L;
        const char* sW = base + OFF_W;
#pragma unroll
        for (int s = 0; s < C::BK / 16; ++s) {
            const int c = s * 2 + (lane >> 5);
            frag bf[C::TN], ah[C::TM], al[C::TM];
#if GEMM_ABLATE == 2
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) asm volatile("" : "=v"(bf[tn]));
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) { asm volatile("" : "=v"(ah[tm])); asm volatile("" : "=v"(al[tm])); }
#else
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) bf[tn] = *(const frag*)(sW + C::off((wn * C::TN + tn) * 32 + (lane & 31), c));
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) {
                ah[tm] = *(const frag*)(sA + C::off((wm * C::TM + tm) * 32 + (lane & 31), c));
                if (SPLIT) al[tm] = *(const frag*)(sL + C::off((wm * C::TM + tm) * 32 + (lane & 31), c));
            }
#endif
#if GEMM_ABLATE == 3
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) asm volatile("" ::"v"(bf[tn]));
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) { asm volatile("" ::"v"(ah[tm])); if (SPLIT) asm volatile("" ::"v"(al[tm])); }
            continue;
#endif
            // (measured: hoisting all fragment reads of a K-tile ahead of the MFMAs, or s_setprio around the MFMA
            //  cluster, is 5-8 % SLOWER on this 1-barrier structure; the compiler's own interleave is kept)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) {
                    acc[tm][tn] = Mfma<T>::run(ah[tm], bf[tn], acc[tm][tn]);
                    if (SPLIT) acc[tm][tn] = Mfma<T>::run(al[tm], bf[tn], acc[tm][tn]);
                }
        }
    }

    gemm_epilogue<T, SPLIT, EPI, C>(p, acc, m0, n0, wm, wn, lane, bz);
}

template <typename T, bool SPLIT, int EPI, typename C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];           // NSTAGE stages
    int base, count;
    xcd_band(p.tiles_m * p.tiles_n, blockIdx.x & 7, base, count);
    gemm_tile<T, SPLIT, EPI, C>(p, base + (blockIdx.x >> 3), smem);
}

// Persistent form for large grids.  PMC on MI355X (profiles/r01_pmc_gemm.json): with one workgroup per tile the
// L2 hit rate of the prior's M = 65536 GEMMs is 68 % and the memory side moves 6x the algorithmic bytes, because
// workgroups that share an A or W panel start whenever a slot frees up and drift apart along K by more than the
// 4 MiB L2 of their XCD can bridge.  Here exactly `p.slots` workgroups per XCD stay resident and walk that XCD's
// band of tiles in CHUNKS of `slots` neighbouring tiles (8 tile rows x 8 tile columns for the default order),
// with a barrier among the XCD's workgroups between chunks: every chunk starts at K = 0 together and streams its
// shared panels through L2 in lock-step.  The barrier is a monotonically increasing counter per XCD (owned by
// the workspace and advanced by the host per launch: p.sync_base); all workgroups of the grid are co-resident by construction (grid = occupancy x
// CUs) and the spin is bounded anyway: the barrier is a locality aid, never a correctness dependency.
template <typename T, bool SPLIT, int EPI, typename C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm_persist_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int base, count;
    const int nwg = p.tiles_m * p.tiles_n;
    xcd_band(nwg, xcd, base, count);
    const int nchunks = ((nwg >> 3) + ((nwg & 7) ? 1 : 0) + p.slots - 1) / p.slots;      // same for every XCD
    int* cnt = p.sync + xcd * 32;                                                       // one 128-B line per XCD
    for (int ch = 0; ch < nchunks; ++ch) {
        const int local = ch * p.slots + slot;
        if (local < count) gemm_tile<T, SPLIT, EPI, C>(p, base + local, smem);
        if (ch + 1 < nchunks) {
            __syncthreads();                                       // also: everyone is done reading LDS of this tile
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int target = p.sync_base + (ch + 1) * p.slots;
                // bounded spin (~30 ms): the barrier only aligns the chunk starts for L2 locality, results never
                // depend on it, so a workgroup that is (unexpectedly) not co-resident cannot hang the kernel
                for (int it = 0; it < 100000 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0; ++it)
                    __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
    }
}

// Chunk barriers every workgroup of a persistent launch takes (same formula as the kernels' `nchunks - 1`).
static int persist_chunk_barriers(int nwg, int slots) {
    const int nchunks = ((nwg >> 3) + ((nwg & 7) ? 1 : 0) + slots - 1) / slots;
    return nchunks > 0 ? nchunks - 1 : 0;
}

// The per-XCD chunk counters live in a caller-owned workspace (include/llark_hip.h: llark_workspace_create) and are
// MONOTONIC: a launch starts at `base`, every workgroup of an XCD arrives once per chunk barrier, so the host advances
// `base` by slots x barriers after enqueueing -- no memset per launch.  Launches that share a workspace must be ordered
// on one stream; the counters are re-zeroed (stream-ordered) long before the 32-bit base can wrap.
static int ws_begin(llark_workspace* ws, GemmParams& p, hipStream_t s) {
    if (!ws || !ws->counters) return -1;
    if (ws->base > (1 << 30)) {
        if (hipMemsetAsync(ws->counters, 0, LLARK_WS_BYTES, s) != hipSuccess) return -1;
        ws->base = 0;
    }
    p.sync = ws->counters;
    p.sync_base = ws->base;
    return 0;
}
static void ws_end(llark_workspace* ws, int nwg, int slots) { ws->base += slots * persist_chunk_barriers(nwg, slots); }

template <typename T, bool SPLIT, int EPI, typename C>
static int launch_gemm_persist(GemmParams p, hipStream_t s, llark_workspace* ws) {
    constexpr int LDS = C::NSTAGE * ((SPLIT ? 2 : 1) * C::A_BYTES + C::B_BYTES);
    auto kern = gemm_persist_kernel<T, SPLIT, EPI, C>;
    static PerDeviceOnce once;                                     // resident workgroups per CU of THIS instantiation, per device
    if (once.first()) {                                            // (0 = unusable)
        int n = 0;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, C::THREADS, LDS) != hipSuccess) n = 0;
        once.slot() = n;
    }
    const int per_cu = once.slot();
    if (!ws || ws->cus % 8) return -1000;
    const int grid = per_cu * ws->cus;
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    if (grid <= 0 || p.batch > 1 || p.tiles_m * p.tiles_n < 4 * grid) return -1000;      // caller falls back to one workgroup per tile
    p.slots = grid / 8;
    if (ws_begin(ws, p, s)) return -1000;
    ws_end(ws, p.tiles_m * p.tiles_n, p.slots);
    kern<<<dim3(grid), C::THREADS, LDS, s>>>(p);
    return check_launch("gemm_persist");
}

template <typename T, bool SPLIT, int EPI, typename C>
static int launch_gemm(GemmParams p, hipStream_t s) {
    constexpr int LDS = C::NSTAGE * ((SPLIT ? 2 : 1) * C::A_BYTES + C::B_BYTES);
    static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    auto kern = gemm_kernel<T, SPLIT, EPI, C>;
    static PerDeviceOnce once;                  // per device: the attribute belongs to the device's function object
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    kern<<<dim3(p.tiles_m * p.tiles_n, p.batch > 0 ? p.batch : 1), C::THREADS, LDS, s>>>(p);
    return check_launch("gemm");
}

template <typename T, typename C>
static int dispatch_persist(const GemmParams& p, bool split, int epi, hipStream_t s, llark_workspace* ws) {
#define CASE(E)                                                      \
    case E:                                                          \
        return split ? launch_gemm_persist<T, true, E, C>(p, s, ws) : launch_gemm_persist<T, false, E, C>(p, s, ws);
    switch (epi) {
        CASE(EPI_F32)
        CASE(EPI_RESID)
        CASE(EPI_QGELU_SPLIT)
        CASE(EPI_OUT16)
        CASE(EPI_SWIGLU16)
        CASE(EPI_SPLIT16)
        CASE(EPI_SWIGLU_SPLIT)
    }
#undef CASE
    return -1000;
}

template <typename T, typename C>
static int dispatch(const GemmParams& p, bool split, int epi, hipStream_t s) {
#define CASE(E)                                                      \
    case E:                                                          \
        return split ? launch_gemm<T, true, E, C>(p, s) : launch_gemm<T, false, E, C>(p, s);
    switch (epi) {
        CASE(EPI_F32)
        CASE(EPI_RESID)
        CASE(EPI_QGELU_SPLIT)
        CASE(EPI_OUT16)
        CASE(EPI_SWIGLU16)
        CASE(EPI_SPLIT16)
        CASE(EPI_SWIGLU_SPLIT)
    }
#undef CASE
    set_error("gemm: unknown epilogue %d", epi);
    return LLARK_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------
// "B-direct" main loop.  Measured on MI355X (profiles/r01_gemm_ablation.txt): the stage-load latency of the
// LDS-staged kernel (~2 us per 64 KiB burst) cannot be hidden inside 160 KiB of LDS.  Here the WEIGHT operand
// never touches LDS: it is pre-packed fragment-major (llark_pack_weight16_frag: one contiguous 1 KiB chunk per
// (32 weight rows, 16 k) = exactly one wave-wide MFMA B fragment), and every wave streams the fragments of its
// own TN column tiles L2 -> VGPR with perfectly coalesced global_load_dwordx4 through a 4-deep register ring
// (3 k16 sub-steps ahead).  Waves are laid out 1 x NW over N, so no weight byte is fetched twice by a block.
// Only A (hi + lo planes) goes through LDS, double-buffered: the tile for K-step kt+1 is loaded into registers
// at the top of K-step kt, stays in flight during the whole compute of kt and is written to the other LDS stage
// just before the single barrier of the K-step.
// ------------------------------------------------------------------------------------------
template <typename T, bool SPLIT, int EPI, typename C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm_bd_kernel(const GemmParams p) {
    typedef typename Mfma<T>::frag frag;
    static_assert(C::WM == 1 && C::BK == 64, "B-direct layout: waves side by side over N, K-step 64");
    constexpr int ASTAGE = (SPLIT ? 2 : 1) * C::A_BYTES;
    constexpr int OFF_L = C::A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 A stages

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = 0, wn = w;

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    constexpr int GM = 8;
    const int gsz = GM * p.tiles_n;
    const int g = bid / gsz;
    const int first_m = g * GM;
    const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
    const int tile_m = first_m + (bid % gsz) % gm;
    const int tile_n = (bid % gsz) / gm;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
    const long long bz = 0;

    const T* Ahi = (const T*)p.Ahi;
    const T* Alo = SPLIT ? (const T*)p.Alo : nullptr;

    // A tile: global -> VGPR -> LDS (ds_write_b128 at the swizzled offset).  Plain loads, so the compiler's
    // counted vmcnt tracking is exact (LDS-DMA would make it drain vmcnt(0) before every ds_read).
    constexpr int APW = (C::BM / C::RPI) / C::NW;                    // 1-KiB row groups per wave and plane
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 areg[(SPLIT ? 2 : 1) * APW];
    const int a_rl = lane / C::CH, a_ch = lane % C::CH;
    auto loadA = [&](int kt) {
        const int k0 = kt * C::BK;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            int r = m0 + (w + i * C::NW) * C::RPI + a_rl;
            r = r < p.M ? r : p.M - 1;
            areg[i] = *(const u32x4*)(Ahi + (size_t)r * p.lda + k0 + a_ch * 8);
            if (SPLIT) areg[APW + i] = *(const u32x4*)(Alo + (size_t)r * p.lda + k0 + a_ch * 8);
        }
    };
    auto storeA = [&](int buf) {
        char* base = smem + buf * ASTAGE;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int row = (w + i * C::NW) * C::RPI + a_rl;
            *(u32x4*)(base + C::off(row, a_ch)) = areg[i];
            if (SPLIT) *(u32x4*)(base + OFF_L + C::off(row, a_ch)) = areg[APW + i];
        }
    };

    // fragment-major weights: chunk (row tile R, k16 step q) at ((R * nk16 + q) * 64 + lane) * 16 B
    const int nk = p.Kp / C::BK;
    const int nk16 = nk * 4;
    const int rtiles = (p.N + 31) >> 5;
    const frag* wbase[C::TN];
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        int R = (n0 >> 5) + wn * C::TN + tn;
        R = R < rtiles ? R : rtiles - 1;                                   // edge tiles: any valid chunk (stores are masked)
        wbase[tn] = (const frag*)p.Wt + ((size_t)R * nk16) * 64 + lane;
    }
    frag ring[4][C::TN];
    auto loadB = [&](int slot, int q) {
        q = q < nk16 ? q : nk16 - 1;                                       // tail: harmless re-load of the last chunk
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) ring[slot][tn] = wbase[tn][(size_t)q * 64];
    };

    f32x16_t acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    loadA(0);
    loadB(0, 0);
    loadB(1, 1);
    loadB(2, 2);
    storeA(0);
    __syncthreads();
    auto load_b1 = [&](int slot, int tn, int q) __attribute__((always_inline)) {
        q = q < nk16 ? q : nk16 - 1;                                       // tail: harmless re-load of the last chunk
        ring[slot][tn] = wbase[tn][(size_t)q * 64];
    };
    auto kloop = [&](auto ntm_tag) __attribute__((always_inline)) {
        constexpr int NTM = decltype(ntm_tag)::value;
        for (int kt = 0; kt < nk; ++kt) {
            loadA(kt + 1 < nk ? kt + 1 : kt);          // unconditional (tail: harmless re-load) so the counted vmcnt waits stay exact
            __builtin_amdgcn_sched_barrier(0);
            const char* sA = smem + (kt & 1) * ASTAGE;
            bd_kstep<T, SPLIT, C, NTM>(sA, sA + OFF_L, lane, ring, acc, kt * 4, load_b1);
            if (kt + 1 < nk) storeA((kt + 1) & 1);
            __syncthreads();
        }
    };
    if (GEMM_BD_TAIL_SKIP && C::TM > 1 && p.M - m0 <= 32) kloop(std::integral_constant<int, 1>{});      // ragged last row tile: one live row block
    else kloop(std::integral_constant<int, C::TM>{});
    if constexpr (EPI == EPI_ROPE_QKV) gemm_epilogue_rope_qkv<T, SPLIT, C>(p, acc, m0, n0, wn, lane, smem);
    else gemm_epilogue<T, SPLIT, EPI, C>(p, acc, m0, n0, wm, wn, lane, bz);
}

template <typename T, bool SPLIT, int EPI, typename C>
static int launch_gemm_bd(GemmParams p, hipStream_t s) {
    constexpr int LDS = 2 * (SPLIT ? 2 : 1) * C::A_BYTES;
    auto kern = gemm_bd_kernel<T, SPLIT, EPI, C>;
    static PerDeviceOnce once;                  // per device: the attribute belongs to the device's function object
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    kern<<<dim3(p.tiles_m * p.tiles_n), C::THREADS, LDS, s>>>(p);
    return check_launch("gemm_bd");
}

template <typename T, typename C>
static int dispatch_bd(const GemmParams& p, bool split, int epi, hipStream_t s) {
    if constexpr (GEMM_BDA && std::is_same<T, bf16_t>::value && C::BM == 128 && C::BN == 256) {
        if (split == (p.Alo != nullptr) && (split || GEMM_BDA >= 2)) {
            const int rc = launch_gemm_bda(p, LLARK_BF16, epi, s);
            if (rc != -1000) return rc;
        }
    }
#define CASE(E)                                                      \
    case E:                                                          \
        return split ? launch_gemm_bd<T, true, E, C>(p, s) : launch_gemm_bd<T, false, E, C>(p, s);
    switch (epi) {
        CASE(EPI_F32)
        CASE(EPI_RESID)
        CASE(EPI_QGELU_SPLIT)
        CASE(EPI_OUT16)
        CASE(EPI_SPLIT16)
    }
    if constexpr (C::TN >= 2) {                                       // SwiGLU pairs (gate, up) live in one wave's tiles
        switch (epi) {
            CASE(EPI_SWIGLU16)
            CASE(EPI_SWIGLU_SPLIT)
        }
    }
#undef CASE
    set_error("gemm: unknown epilogue %d", epi);
    return LLARK_ERR_INVALID;
}

// [N][ld] row-major 16-bit weights -> fragment-major chunks (see gemm_bd_kernel); rows >= n are zero.
__global__ void pack_frag_kernel(const unsigned short* __restrict__ src, int ld, int n, int kp, uint4* __restrict__ dst, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-B lane slot each
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const long long chunk = i >> 6;
    const int nk16 = kp >> 4;
    const int q = (int)(chunk % nk16);
    const int R = (int)(chunk / nk16);
    const int row = R * 32 + (lane & 31);
    const int k = q * 16 + (lane >> 5) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < n) v = *(const uint4*)(src + (size_t)row * ld + k);
    dst[i] = v;
}

// ------------------------------------------------------------------------------------------
// Skinny GEMM for decode (M <= 16 rows: one new token per sequence): HBM-bound weight streaming.
// Block = 4 waves owning 16 output columns (x NT tiles); wave w streams the K-slice [w*K/4, (w+1)*K/4) of
// the weight rows straight from HBM into MFMA B fragments (no LDS: nothing is shared between waves), the
// 16 x K activation slab comes from L2; partial 16x16 tiles are reduced through LDS by wave 0, which also
// applies the epilogue.  v_mfma_f32_16x16x32 keeps the fp32 accumulation / hi+lo semantics of the big kernel.
// ------------------------------------------------------------------------------------------
template <typename T>
struct Mfma16;
template <>
struct Mfma16<half_t> {
    static __device__ __forceinline__ f32x4_t run(half8_t a, half8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <>
struct Mfma16<bf16_t> {
    static __device__ __forceinline__ f32x4_t run(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

template <typename T, bool SPLIT, int EPI, int KW>
__global__ __launch_bounds__(KW * 64) void gemm_skinny_kernel(const GemmParams p) {
    typedef typename Mfma<T>::frag frag;
    constexpr int NT = IS_SWIGLU(EPI) ? 2 : 1;
    __shared__ float red[KW][NT][16][17];
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // column tiles: plain: rows n0..n0+15 of W.  SwiGLU: gate rows 64q+16t.., up rows 64q+32+16t.. (t = 0,1)
    int wrow[NT];
    int ocol;
    if (IS_SWIGLU(EPI)) {
        const int q = blockIdx.x >> 1, t = blockIdx.x & 1;
        wrow[0] = 64 * q + 16 * t;
        wrow[NT - 1] = 64 * q + 32 + 16 * t;
        ocol = 32 * q + 16 * t;
    } else {
        wrow[0] = blockIdx.x * 16;
        ocol = wrow[0];
    }
    const int ksteps = p.Kp / 32;
    const int per = (ksteps + KW - 1) / KW;
    const int ks0 = w * per < ksteps ? w * per : ksteps, ks1 = (ks0 + per) < ksteps ? (ks0 + per) : ksteps;
    const T* Ahi = (const T*)p.Ahi;
    const T* Alo = (const T*)p.Alo;
    const T* Wt = (const T*)p.Wt;
    const int arow = c < p.M ? c : p.M - 1;
    const bool a_live = c < p.M;
    const T* ah_p = Ahi ? Ahi + (size_t)arow * p.lda + g * 8 : nullptr;
    const T* al_p = (SPLIT && Alo) ? Alo + (size_t)arow * p.lda + g * 8 : nullptr;
    const T* w_p[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int r = wrow[t] + c;
        r = r < p.N ? r : p.N - 1;
        w_p[t] = Wt + (size_t)r * p.ldw + g * 8;
    }
    // Fused RMSNorm of the A rows (decode): every workgroup re-derives the <= 16 row scales itself -- one wave per row,
    // the per-lane float4 order and wave reduction of rmsnorm_kernel (llama.hip), so rstd and therefore the hi / lo
    // operand planes are bit-identical to a separate rmsnorm launch -- and normalises its A fragments on the fly:
    // y = g * (x * rstd), hi = bf16(y), lo = bf16(y - hi).  Costs ~128 KiB of L2 reads per workgroup, saves a launch.
    __shared__ float s_rstd[16];
    const bool norm_a = p.xn != nullptr;
    float my_rstd = 0.0f;
    const float* xn_p = nullptr;
    if (norm_a) {
        const int w4 = p.Kp >> 2;
        for (int row = w; row < p.M; row += KW) {
            const float4* xr = (const float4*)(p.xn + (size_t)row * p.ldxn);
            float sq = 0.0f;
            for (int cidx = lane; cidx < w4; cidx += 64) {
                const float4 v = xr[cidx];
                sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            const float var = wave_sum(sq) / (float)p.Kp;
            if (lane == 0) s_rstd[row] = 1.0f / sqrtf(var + p.xeps);
        }
        __syncthreads();
        my_rstd = s_rstd[arow];
        xn_p = p.xn + (size_t)arow * p.ldxn + g * 8;
    }
    auto load_a = [&](int kstep, frag& hi, frag& lo) __attribute__((always_inline)) {
        if (!norm_a) {
            // Rows >= M of the 16-row MFMA tile are never stored: their lanes load nothing (zero fragments) instead of
            // re-reading row M-1 -- at M = 1 that was 16 copies of the same activation row per k-step through L2.
            if (a_live) {
                hi = *(const frag*)(ah_p + kstep * 32);
                if (SPLIT) lo = *(const frag*)(al_p + kstep * 32);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    hi[e] = (T)0.0f;
                    if (SPLIT) lo[e] = (T)0.0f;
                }
            }
        } else {
            const float4 x0 = *(const float4*)(xn_p + kstep * 32), x1 = *(const float4*)(xn_p + kstep * 32 + 4);
            const float4 g0 = *(const float4*)(p.xg + kstep * 32 + g * 8), g1 = *(const float4*)(p.xg + kstep * 32 + g * 8 + 4);
            const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = gv[e] * (xv[e] * my_rstd);
                const T h = Mfma<T>::cvt(y);
                hi[e] = h;
                if (SPLIT) lo[e] = Mfma<T>::cvt(y - Mfma<T>::back(h));
            }
        }
    };
    f32x4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // residual values of the epilogue are requested now, not after the K loop: at M <= 16 the kernel is a chain of memory
    // latencies and this one can hide behind the weight stream (R may alias C: nothing has been stored yet)
    float rpre[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_RESID && w == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * g + r, n = ocol + c;
            if (m < p.M && n < p.N) rpre[r] = p.R[(size_t)m * p.ldr + n];
        }
    }
#ifndef SKINNY_DEPTH
#define SKINNY_DEPTH 4
#endif
#ifndef SKINNY_NT
#define SKINNY_NT 0
#endif
    constexpr int DEPTH = SKINNY_DEPTH;
    auto load_w = [&](const T* ptr) __attribute__((always_inline)) {
#if SKINNY_NT
        return __builtin_nontemporal_load((const frag*)ptr);        // streamed once: do not keep the line
#else
        return *(const frag*)ptr;
#endif
    };
    int ks = ks0;
    for (; ks + DEPTH <= ks1; ks += DEPTH) {            // DEPTH k-steps in flight: DEPTH x 64 contiguous bytes per weight row
        frag bw[NT][DEPTH], ah[DEPTH], al[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bw[t][u] = load_w(w_p[t] + (ks + u) * 32);
            load_a(ks + u, ah[u], al[u]);
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = Mfma16<T>::run(ah[u], bw[t][u], acc[t]);
                if (SPLIT) acc[t] = Mfma16<T>::run(al[u], bw[t][u], acc[t]);
            }
    }
    for (; ks < ks1; ++ks) {
        frag a1, a1l;
        load_a(ks, a1, a1l);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const frag bw = load_w(w_p[t] + ks * 32);
            acc[t] = Mfma16<T>::run(a1, bw, acc[t]);
            if (SPLIT) acc[t] = Mfma16<T>::run(a1l, bw, acc[t]);
        }
    }
    // C layout of 16x16 MFMA: col = c (weight row within the tile), rows m = 4g + r
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][t][4 * g + r][c] = acc[t][r];
    __syncthreads();
    if (w == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * g + r;
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float part[KW];
#pragma unroll
            for (int q = 0; q < KW; ++q) part[q] = red[q][t][m][c];
#pragma unroll
            for (int st = 1; st < KW; st *= 2)                  // fixed pairwise tree: deterministic
#pragma unroll
                for (int q = 0; q < KW; q += 2 * st) part[q] += part[q + st];
            v[t] = part[0];
        }
        const int n = ocol + c;
        const int nlim = IS_SWIGLU(EPI) ? (p.N >> 1) : p.N;
        if (m >= p.M || n >= nlim) continue;
        if (EPI == EPI_F32) {
            p.C[(size_t)m * p.ldc + n] = v[0] + (p.bias ? p.bias[n] : 0.0f);
        } else if (EPI == EPI_RESID) {
            const float hv = rpre[r] + (v[0] + (p.bias ? p.bias[n] : 0.0f));
            // fused-norm mode: write-through (sc1) stores, so that the last workgroup can read the rows without any release fence here
            if (p.nw != nullptr) __hip_atomic_store(p.C + (size_t)m * p.ldc + n, hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else p.C[(size_t)m * p.ldc + n] = hv;
        } else if (IS_SWIGLU(EPI)) {
            const float a = silu(v[0]) * v[NT - 1];
            const T hi = Mfma<T>::cvt(a);
            ((T*)p.Ohi)[(size_t)m * p.ldo + n] = hi;
            if (EPI == EPI_SWIGLU_SPLIT) ((T*)p.Olo)[(size_t)m * p.ldo + n] = Mfma<T>::cvt(a - Mfma<T>::back(hi));
        }
    }
    }
    if (EPI == EPI_RESID && p.nw != nullptr) {
        // Fused RMSNorm (decode): the workgroup that arrives last owns complete rows of C = h and normalises them
        // with the arithmetic of rmsnorm_kernel (llama.hip) -- same per-lane float4 order, same wave reduction,
        // y = g * (x * rstd), hi = bf16(y), lo = bf16(y - hi) -- so the result is bit-identical to a separate launch.
        // Hand-off (publish / consume recipe of the CDNA guide): the C stores above are write-through, the storing wave drains
        // them, ONE lane takes a ticket with a relaxed agent-scope atomic; the last workgroup does ONE agent-scope acquire (drops
        // its CU's stale L1 lines) and reads the rows with plain loads.  (The first version fenced with __threadfence() in every
        // workgroup -- an L2 write-back + invalidate each, ~65 us per launch on the 8-XCD part.)
        __shared__ int s_last;
        if (w == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int last = __hip_atomic_fetch_add(p.ncnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
            if (last) {
                __hip_atomic_store(p.ncnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            s_last = last;
        }
        __syncthreads();
        if (s_last) {
            const int w4 = p.N >> 2;
            for (int row = w; row < p.M; row += KW) {
                const float4* xr = (const float4*)(p.C + (size_t)row * p.ldc);
                float sq = 0.0f;
                for (int cidx = lane; cidx < w4; cidx += 64) {
                    const float4 v = xr[cidx];
                    sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
                const float var = wave_sum(sq) / (float)p.N;
                const float rstd = 1.0f / sqrtf(var + p.neps);
                const float4* g4 = (const float4*)p.nw;
                bf16x4_t* hr = (bf16x4_t*)((bf16_t*)p.nhi + (size_t)row * p.ldn);
                bf16x4_t* lr = p.nlo ? (bf16x4_t*)((bf16_t*)p.nlo + (size_t)row * p.ldn) : nullptr;
                for (int cidx = lane; cidx < w4; cidx += 64) {
                    const float4 v = xr[cidx];
                    const float4 gg = g4[cidx];
                    float y[4] = {gg.x * (v.x * rstd), gg.y * (v.y * rstd), gg.z * (v.z * rstd), gg.w * (v.w * rstd)};
                    bf16x4_t hh, ll;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hh[e] = (bf16_t)y[e];
                        ll[e] = (bf16_t)(y[e] - (float)hh[e]);
                    }
                    hr[cidx] = hh;
                    if (lr) lr[cidx] = ll;
                }
            }
        }
    }
}

template <typename T, bool SPLIT, int EPI>
static int launch_skinny(const GemmParams& p, hipStream_t s) {
    const int blocks = IS_SWIGLU(EPI) ? (p.N / 64) * 2 : cdiv(p.N, 16);
    // K is split over the waves of a block; HBM streaming needs many bytes in flight per CU, so problems with few
    // column tiles (o_proj / down_proj: 256 blocks) or a deep K get 16 waves, the wide ones 8
    const int kw = (blocks <= 512 || p.Kp >= 8192) ? 16 : 8;
    if (kw == 16) gemm_skinny_kernel<T, SPLIT, EPI, 16><<<blocks, 1024, 0, s>>>(p);
    else if (kw == 8) gemm_skinny_kernel<T, SPLIT, EPI, 8><<<blocks, 512, 0, s>>>(p);
    else gemm_skinny_kernel<T, SPLIT, EPI, 4><<<blocks, 256, 0, s>>>(p);
    return check_launch("gemm_skinny");
}

template <typename T>
static int dispatch_skinny(const GemmParams& p, bool split, int epi, hipStream_t s) {
#define SK(E)                                                                                             \
    case E:                                                                                               \
        return split ? launch_skinny<T, true, E>(p, s) : launch_skinny<T, false, E>(p, s);
    switch (epi) {
        SK(EPI_F32)
        SK(EPI_RESID)
        SK(EPI_SWIGLU16)
        SK(EPI_SWIGLU_SPLIT)
    }
#undef SK
    return 1;      // not a skinny epilogue: caller falls back to the tiled kernel
}

// Tile variants (tuning knob; every variant computes the same result).
typedef Cfg<2, 2, 2, 2, 32, 3> Cfg0;   // 128x128x32, 4 waves, 48 KiB (split)      : 3 blocks/CU
typedef Cfg<4, 2, 2, 2, 32, 4> Cfg1;   // 256x128x32, 8 waves, 80 KiB               : 2 blocks/CU
typedef Cfg<2, 2, 2, 4, 32, 2> Cfg2;   // 128x256x32, 4 waves (64x128 per wave), 64 KiB : 2 blocks/CU
// BK = 64: every DMA instruction moves 8 rows x 128 B (full cache lines) instead of 16 rows x 64 B
typedef Cfg<2, 2, 2, 2, 64, 3, 1> Cfg11;  // 128x128x64, 4 waves, single stage 48 KiB                    : 3 blocks/CU
typedef Cfg<2, 2, 4, 4, 64, 1, 2> Cfg13;  // 256x256x64, 4 waves (128x128 per wave), two stages 128 KiB (plain only) : 1 block/CU
typedef Cfg<2, 2, 2, 4, 64, 2, 1> Cfg12;  // 128x256x64, 4 waves (64x128 per wave), single stage 64 KiB  : 2 blocks/CU
// B-direct kernels (weights fragment-major, L2 -> VGPR): waves 1 x 4 over N

template <typename T>
static int dispatch_variant(int variant, const GemmParams& p, bool split, int epi, hipStream_t s, llark_workspace* ws) {
    switch (variant) {
        case 0: return dispatch<T, Cfg0>(p, split, epi, s);
        case 1: return dispatch<T, Cfg1>(p, split, epi, s);
        case 2: return dispatch<T, Cfg2>(p, split, epi, s);
        case 11: return dispatch<T, Cfg11>(p, split, epi, s);
        case 12: return dispatch<T, Cfg12>(p, split, epi, s);
        case 13:
            if (split) break;
            switch (epi) {
                case EPI_F32: return launch_gemm<T, false, EPI_F32, Cfg13>(p, s);
                case EPI_RESID: return launch_gemm<T, false, EPI_RESID, Cfg13>(p, s);
                case EPI_OUT16: return launch_gemm<T, false, EPI_OUT16, Cfg13>(p, s);
                case EPI_SWIGLU16: return launch_gemm<T, false, EPI_SWIGLU16, Cfg13>(p, s);
            }
            break;
        case 32: {                                                  // gemm256n's tile on the 16x16x32 instruction (gemm256x.hip), else 31
            if (split && ws && ws->cus % 8 == 0) {
                GemmParams q = p;
                const int dt = std::is_same<T, half_t>::value ? LLARK_F16 : LLARK_BF16;
                if (ws_begin(ws, q, s) == 0) {
                    const int rc = launch_gemm256x(q, dt, epi, s, ws->cus);
                    if (rc != -1000) {
                        ws_end(ws, cdiv(p.M, 256) * cdiv(p.N, 256), ws->cus / 8);
                        return rc;
                    }
                }
            }
        }
        [[fallthrough]];
        case 31: {                                                  // 256x256x64 split tile, phases over N with resident A fragments (gemm256n.hip), else 30
            if (split && ws && ws->cus % 8 == 0) {
                GemmParams q = p;
                const int dt = std::is_same<T, half_t>::value ? LLARK_F16 : LLARK_BF16;
                if (ws_begin(ws, q, s) == 0) {
                    const int rc = launch_gemm256n(q, dt, epi, s, ws->cus);
                    if (rc != -1000) {
                        ws_end(ws, cdiv(p.M, 256) * cdiv(p.N, 256), ws->cus / 8);
                        return rc;
                    }
                }
            }
        }
        [[fallthrough]];
        case 30: {                                                  // 256x256x64 split tile with the counted-vmcnt LDS ring (gemm256.hip), else 20
            if (split && ws && ws->cus % 8 == 0) {
                GemmParams q = p;
                const int dt = std::is_same<T, half_t>::value ? LLARK_F16 : LLARK_BF16;
                if (ws_begin(ws, q, s) == 0) {
                    const int rc = launch_gemm256(q, dt, epi, s, ws->cus);
                    if (rc != -1000) {
                        ws_end(ws, cdiv(p.M, 256) * cdiv(p.N, 256), ws->cus / 8);
                        return rc;
                    }
                }
            }
        }
        [[fallthrough]];
        case 20: {                                                  // persistent 128x256x64 (large grids), else plain variant 12
            const int rc = dispatch_persist<T, Cfg12>(p, split, epi, s, ws);
            return rc == -1000 ? dispatch<T, Cfg12>(p, split, epi, s) : rc;
        }
    }
    set_error("gemm: unknown tile variant %d", variant);
    return LLARK_ERR_INVALID;
}

}  // namespace llark

using namespace llark;

// dtype: LLARK_F16 / LLARK_BF16.  split != 0 -> A is given as hi/lo planes (Alo required).
// Wt is [N][ldw] (K-contiguous rows, i.e. the transpose of upstream Conv1D.w / the native layout of
// nn.Linear.weight); K is padded with zeros up to kp (multiple of 32) in BOTH A and Wt.  M and N
// need no padding: out-of-range rows are clamped on load and masked on store.
// Default tile choice, from the MI355X sweeps in profiles/r01_gemm_variants*.txt.  An ablation of the main
// loop (profiles/r01_gemm_ablation.txt) shows the L2->LDS DMA stream, not the MFMAs, is the long pole, so the
// winners are the shapes that move the fewest bytes per flop in FULL 128-B lines: BK = 64 (8 rows x 128 B per
// DMA instruction), 64x128 per wave, single LDS stage with 2 blocks per CU overlapping each other.
static int pick_variant(int split, int m, int n, int kp, bool has_ws) {
    if (m <= 128) return 0;
    // plain 16-bit products: 128x128x64 tiles at 3 workgroups per CU win or tie on every HTSAT linear (M = 4096 .. 262144,
    // N = 128 .. 4096, K = 384 .. 12288; profiles/r01_clap_gemm_sweep.txt: 8.8 ms per forward vs 9.8 ms with the rules
    // below, which were swept on the split-mode prior shapes) as they already did on the Llama prefill shapes.
    if (!split && kp % 64 == 0 && n < 16384) return 11;
    if (n < 256) return 0;
    if (kp % 64 != 0) return kp < 2048 ? 1 : 2;
    const bool persist = has_ws;                  // the persistent kernels keep their chunk counters in the caller's workspace
    // the prior (M = clips x 8192, split fp16): 256x256x64 tile, phases over N with resident A fragments (gemm256n.hip; round 2's
    // M-split LDS ring gemm256.hip = variant 30 stays for A/B) as soon as the
    // problem has two tiles per CU -- also at B = 1 (608 tiles), so a clip's result does not depend on the batch it rides in
    // round 4: the same tile on v_mfma_f32_16x16x32 (gemm256x.hip; falls back to 31 for the shapes / epilogues it does not take) for the
    // deep products: the instruction costs 8.6 % more pipe cycles and buys a 16 % higher sustained clock (1.44 -> 1.68 GHz under the
    // profiler, profiles/r04_pmc_gemm256x_vs_n.txt): +4 .. 5 % at K = 4800, but -3 % at K = 1216, where a quarter of the tile time is
    // epilogue and the 32x32x16 loop is not power-bound to begin with (profiles/r04_gemm256x_ab_v2.txt)
    // (from 384 tiles = 1.5 per CU on: one clip's c_attn has 32 x 15 = 480, and a clip's result must not depend on the batch it rides in --
    //  variants 31 and 12 are bit-identical to each other, variant 32 is not)
    if (split && persist && kp >= 128 && (long)cdiv(m, 256) * cdiv(n, 256) >= 384) return kp >= 2048 ? 32 : 31;
    if (kp < 2048) return 12;                     // shallow K (attention c_proj, K = 1216): per-tile 128x256x64 (the chunk barrier of the
                                                  // persistent form does not pay off over 19 K-steps); re-swept after the residual-epilogue fix
    if (m >= 16384 && persist) return 20;                    // very tall products (M = 65536): persistent, chunk-synchronous (L2 hit rate 68 -> 83 %)
    if (!split && n < 16384) return 11;           // plain 16-bit, mid-size N (Llama q/k/v/o, down): 128x128x64, 3 blocks/CU
    return 12;                                    // 128x256x64
}

static int gemm16_impl(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                       const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c, int ldc,
                       const float* resid, int ldr, void* out_hi, void* out_lo, int ldo, int batch, long long sa,
                       long long sw, long long sc, long long sr, long long so, llark_stream_t stream, void* out_hi2 = nullptr,
                       int act = 0, llark_workspace* ws = nullptr) {
    LLARK_REQUIRE(a_hi && wt && m > 0 && n > 0 && kp > 0, "gemm16: null pointer or empty problem");
    LLARK_REQUIRE(kp % 64 == 0 || (kp % 32 == 0 && variant < 10),
                  "gemm16: kp=%d must be a multiple of the K-step (zero-pad K)", kp);
    LLARK_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= kp && ldw >= kp, "gemm16: lda/ldw must be >= kp and multiples of 8");
    LLARK_REQUIRE(!split || a_lo, "gemm16: split mode needs the lo plane");
    LLARK_REQUIRE(((uintptr_t)a_hi & 15) == 0 && ((uintptr_t)wt & 15) == 0 && (!a_lo || ((uintptr_t)a_lo & 15) == 0),
                  "gemm16: operands must be 16-byte aligned");
    if (epilogue == EPI_F32 || epilogue == EPI_RESID) LLARK_REQUIRE(c && ldc >= n, "gemm16: fp32 output missing");
    if (epilogue == EPI_RESID) LLARK_REQUIRE(resid && ldr >= n, "gemm16: residual missing");
    if (epilogue == EPI_QGELU_SPLIT || epilogue == EPI_SPLIT16) LLARK_REQUIRE(out_hi && out_lo && ldo >= n, "gemm16: split outputs missing");
    if (epilogue == EPI_OUT16) LLARK_REQUIRE(out_hi && ldo >= n, "gemm16: 16-bit output missing");
    if (IS_SWIGLU(epilogue)) LLARK_REQUIRE(out_hi && n % 64 == 0 && ldo >= n / 2, "gemm16: swiglu needs n%%64==0 and an output");
    if (epilogue == EPI_SWIGLU_SPLIT) LLARK_REQUIRE(out_lo, "gemm16: swiglu-split needs the lo output");
    GemmParams p = {};
    p.Ahi = a_hi; p.Alo = a_lo; p.lda = lda; p.Wt = wt; p.ldw = ldw; p.bias = bias;
    p.M = m; p.N = n; p.Kp = kp; p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr;
    p.Ohi = out_hi; p.Olo = out_lo; p.ldo = ldo; p.Ohi2 = out_hi2; p.act = act;
    p.tiles_m = p.tiles_n = 0;
    p.batch = batch; p.sA = sa; p.sW = sw; p.sC = sc; p.sR = sr; p.sO = so;
    hipStream_t s = (hipStream_t)stream;
    if (variant < 0 && m <= 16 && batch <= 1 && act == 0 && !out_hi2) {   // decode: HBM-bound skinny kernel
        int rc = 1;
        if (dtype == LLARK_F16) rc = dispatch_skinny<half_t>(p, split != 0, epilogue, s);
        else if (dtype == LLARK_BF16) rc = dispatch_skinny<bf16_t>(p, split != 0, epilogue, s);
        if (rc != 1) return rc;
    }
    if (variant < 0) variant = pick_variant(split, m, n, kp, ws != nullptr);
    if (dtype == LLARK_F16) return dispatch_variant<half_t>(variant, p, split != 0, epilogue, s, ws);
    if (dtype == LLARK_BF16) return dispatch_variant<bf16_t>(variant, p, split != 0, epilogue, s, ws);
    set_error("gemm16: unknown dtype %d", dtype);
    return LLARK_ERR_INVALID;
}

extern "C" int llark_gemm16_ex(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                               const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c, int ldc,
                               const float* resid, int ldr, void* out_hi, void* out_lo, int ldo,
                               llark_stream_t stream) {
    return gemm16_impl(variant, dtype, split, epilogue, a_hi, a_lo, lda, wt, ldw, bias, m, n, kp, c, ldc, resid, ldr, out_hi,
                       out_lo, ldo, 0, 0, 0, 0, 0, 0, stream);
}

// llark_gemm16_ex with a caller-owned workspace: enables the persistent, chunk-synchronous tile variants (20, 30), whose
// per-XCD chunk counters live in the workspace.  Without one (llark_gemm16 / llark_gemm16_ex) those variants are never
// chosen and a request for them runs the per-tile 128x256x64 kernel: no entry point allocates or keeps hidden state.
extern "C" int llark_gemm16_ws(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                               const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c, int ldc,
                               const float* resid, int ldr, void* out_hi, void* out_lo, int ldo, llark_workspace_t ws,
                               llark_stream_t stream) {
    return gemm16_impl(variant, dtype, split, epilogue, a_hi, a_lo, lda, wt, ldw, bias, m, n, kp, c, ldc, resid, ldr, out_hi,
                       out_lo, ldo, 0, 0, 0, 0, 0, 0, stream, nullptr, 0, ws);
}

// LayerNorm folded into the GEMM epilogues around it (round 4; csrc/gemm256x.hip).  One entry point, two roles:
//   consumer: ln_stat != NULL -- a_hi / a_lo are the planes of x . gamma (written by a producer launch), the epilogue applies the row's
//             (mean, rstd): out = rstd (acc - mean ln_vec[n]) + bias[n]; ln_vec[n] = sum_k gamma_k W[n][k], bias[n] = sum_k beta_k W[n][k] + b[n];
//             epilogue LLARK_EPI_F32 (c) or LLARK_EPI_QGELU_SPLIT (out_hi / out_lo)
//   producer: ln_part != NULL -- LLARK_EPI_RESID: c = resid + acc + bias as usual, plus out_hi / out_lo [m][ldo] = hi / lo of c . ln_vec[n]
//             (ln_vec = the NEXT LayerNorm's gamma) and ln_part [m][2 * ceil(n / 256)][2] = per-(row, 128-column slice) sum and sum of
//             squares of c; llark_ln_stats_finalize turns them into [m][2] (mean, rstd)
// Only the 256x256 tile of gemm256x.hip implements the roles: shapes it does not take return LLARK_ERR_UNSUPPORTED before anything is
// launched (ask llark_gemm16_ln_takes first).
extern "C" int llark_gemm16_ln_takes(int m, int n, int kp) { return gemm256x_takes(m, n, kp) ? 1 : 0; }

extern "C" int llark_gemm16_ln(int dtype, int epilogue, const void* a_hi, const void* a_lo, int lda, const void* wt, int ldw, const float* bias, int m,
                               int n, int kp, float* c, int ldc, const float* resid, int ldr, void* out_hi, void* out_lo, int ldo,
                               const float* ln_stat, const float* ln_vec, float* ln_part, llark_workspace_t ws, llark_stream_t stream) {
    return llark_gemm16_ln_p(dtype, epilogue, a_hi, a_lo, lda, wt, ldw, bias, m, n, kp, c, ldc, resid, ldr, out_hi, out_lo, ldo, ln_stat, ln_vec, ln_part,
                             nullptr, ws, stream);
}

// The same two roles; the producer may be given the rows' PREDICTED statistics ln_pred [m][2] = (shift, scale) -- see GemmParams::ln_pred --
// so that the planes it writes are of order one: pair it with llark_ln_stats_finalize_p.  ln_pred = NULL is llark_gemm16_ln.
extern "C" int llark_gemm16_ln_p(int dtype, int epilogue, const void* a_hi, const void* a_lo, int lda, const void* wt, int ldw, const float* bias, int m,
                                 int n, int kp, float* c, int ldc, const float* resid, int ldr, void* out_hi, void* out_lo, int ldo,
                                 const float* ln_stat, const float* ln_vec, float* ln_part, const float* ln_pred, llark_workspace_t ws,
                                 llark_stream_t stream) {
    LLARK_REQUIRE(a_hi && a_lo && wt && ln_vec && ws && m > 0 && n > 0 && kp > 0, "gemm16_ln: null pointer or empty problem");
    LLARK_REQUIRE(ln_pred == nullptr || ln_part != nullptr, "gemm16_ln: ln_pred belongs to the producer role (ln_part)");
    LLARK_REQUIRE((ln_stat != nullptr) != (ln_part != nullptr), "gemm16_ln: exactly one of ln_stat (consumer) / ln_part (producer) must be given");
    LLARK_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= kp && ldw >= kp, "gemm16_ln: lda/ldw must be >= kp and multiples of 8");
    if (ln_part) LLARK_REQUIRE(epilogue == EPI_RESID && c && resid && out_hi && out_lo && ldc >= n && ldr >= n && ldo >= n, "gemm16_ln: the producer role is LLARK_EPI_RESID with c, resid and the operand planes");
    else LLARK_REQUIRE((epilogue == EPI_F32 && c && ldc >= n) || (epilogue == EPI_QGELU_SPLIT && out_hi && out_lo && ldo >= n), "gemm16_ln: the consumer role is LLARK_EPI_F32 or LLARK_EPI_QGELU_SPLIT");
    if (!gemm256x_takes(m, n, kp) || ws->cus % 8) {
        set_error("gemm16_ln: m=%d n=%d kp=%d is not a shape the 256x256 tile takes", m, n, kp);
        return LLARK_ERR_UNSUPPORTED;
    }
    GemmParams p = {};
    p.Ahi = a_hi; p.Alo = a_lo; p.lda = lda; p.Wt = wt; p.ldw = ldw; p.bias = bias;
    p.M = m; p.N = n; p.Kp = kp; p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr;
    p.Ohi = out_hi; p.Olo = out_lo; p.ldo = ldo;
    p.ln_stat = ln_stat; p.ln_vec = ln_vec; p.ln_part = ln_part; p.ln_pred = ln_pred;
    hipStream_t s = (hipStream_t)stream;
    if (ws_begin(ws, p, s)) {
        set_error("gemm16_ln: workspace without counters");
        return LLARK_ERR_INVALID;
    }
    const int rc = launch_gemm256x(p, dtype, epilogue, s, ws->cus);
    if (rc == -1000) {
        set_error("gemm16_ln: operands do not meet the tile's alignment rules");
        return LLARK_ERR_UNSUPPORTED;
    }
    ws_end(ws, cdiv(m, 256) * cdiv(n, 256), ws->cus / 8);
    return rc;
}

// The producer role with fragment-major weights on the DMA loop's 128x256 tiles, two workgroups to a CU (gemm_bda.hip,
// gemm_bda_lnp_kernel): for the products whose K loop is too short to amortise the persistent tile's serial epilogue (the prior's
// attention-output c_proj, K = 1216).  ln_part is [m][ceil(n / 64)][2]: 64-column slices.
extern "C" int llark_gemm16_lnp_fragw(int dtype, const void* a_hi, const void* a_lo, int lda, const void* wfrag, const float* bias, int m, int n, int kp,
                                      float* c, int ldc, const float* resid, int ldr, void* out_hi, void* out_lo, int ldo, const float* ln_vec,
                                      float* ln_part, const float* ln_pred, llark_stream_t stream) {
    LLARK_REQUIRE(a_hi && a_lo && wfrag && ln_vec && ln_part && c && resid && out_hi && out_lo && m > 0 && n > 0 && kp > 0,
                  "gemm16_lnp_fragw: null pointer or empty problem");
    LLARK_REQUIRE(kp % 64 == 0 && lda % 8 == 0 && lda >= kp && ldc >= n && ldr >= n && ldo >= n, "gemm16_lnp_fragw: kp must be a multiple of 64, lda >= kp a multiple of 8, ldc / ldr / ldo >= n");
    LLARK_REQUIRE(((uintptr_t)a_hi & 15) == 0 && ((uintptr_t)a_lo & 15) == 0 && ((uintptr_t)wfrag & 15) == 0, "gemm16_lnp_fragw: operands must be 16-byte aligned");
    GemmParams p = {};
    p.Ahi = a_hi; p.Alo = a_lo; p.lda = lda; p.Wt = wfrag; p.ldw = 0; p.bias = bias; p.M = m; p.N = n; p.Kp = kp;
    p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr; p.Ohi = out_hi; p.Olo = out_lo; p.ldo = ldo;
    p.ln_vec = ln_vec; p.ln_part = ln_part; p.ln_pred = ln_pred;
    const int rc = launch_gemm_bda_lnp(p, dtype, (hipStream_t)stream);
    if (rc == -1000) {
        set_error("gemm16_lnp_fragw: needs f16 / bf16 planes, kp >= 192, n %% 4 == 0, 16-byte aligned fp32 rows (ldc, ldr multiples of 4) and 8-byte aligned plane rows (ldo a multiple of 4)");
        return LLARK_ERR_UNSUPPORTED;
    }
    return rc;
}

extern "C" llark_workspace_t llark_workspace_create(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        set_error("workspace_create: cannot query the current device");
        return nullptr;
    }
    llark_workspace* ws = new (std::nothrow) llark_workspace();
    if (!ws) { set_error("workspace_create: out of host memory"); return nullptr; }
    ws->device = dev; ws->cus = cus; ws->base = 0; ws->counters = nullptr;
#ifdef G256X_PROF      // profiling build only (gemm256x.hip): fewer resident workgroups per XCD, to size the epilogue bursts
    if (const char* e = getenv("LLARK_G256X_CUS")) ws->cus = atoi(e);
#endif
    if (hipMalloc((void**)&ws->counters, LLARK_WS_BYTES) != hipSuccess || hipMemset(ws->counters, 0, LLARK_WS_BYTES) != hipSuccess) {
        set_error("workspace_create: cannot allocate %d bytes of device memory", LLARK_WS_BYTES);
        if (ws->counters) (void)hipFree(ws->counters);
        delete ws;
        return nullptr;
    }
    return ws;
}

extern "C" int llark_workspace_destroy(llark_workspace_t ws) {
    if (!ws) return LLARK_OK;
    if (ws->counters) (void)hipFree(ws->counters);
    delete ws;
    return LLARK_OK;
}

// Split GEMM with an e4m3 low plane (csrc/gemm256_lo8n.hip): the prior's Conv1D products in the opt-in "lo8" mode.
//   a_hi  fp16 [m][lda]            = fp16(a)
//   a_lo8 e4m3 [m][lda8] (bytes)   = fp8(sat((a - a_hi) * 2^sa)), every 64-k block in the slot order of lo8_pos()
//   wt    fp16 [n][ldw]; sw such that max|W| * 2^sw <= 448
//   w8    e4m3 [n][ldw8] (bytes) = fp8(wt * 2^sw) in the same slot order (llark_pack_weight_lo8), staged through LDS next to wt
//         (required; the form that derived it from wt in registers lost: git history, scripts/experiments/ before round 4)
// epilogue: LLARK_EPI_F32 / LLARK_EPI_RESID / LLARK_EPI_QGELU_SPLIT8 (out_hi fp16 [m][ldo], out_lo8 e4m3 [m][ldo8]).
extern "C" int llark_gemm16_lo8(int epilogue, const void* a_hi, const void* a_lo8, int lda, int lda8, const void* wt, int ldw,
                                const void* w8, int ldw8, const float* bias, int m, int n, int kp, int sa, int sw, float* c, int ldc, const float* resid,
                                int ldr, void* out_hi, void* out_lo8, int ldo, int ldo8, llark_workspace_t ws,
                                llark_stream_t stream) {
    LLARK_REQUIRE(a_hi && a_lo8 && wt && w8 && ws && m > 0 && n > 0, "gemm16_lo8: null pointer (a_hi / a_lo8 / wt / w8 / ws) or empty problem");
    LLARK_REQUIRE(kp % 64 == 0 && kp >= 128, "gemm16_lo8: kp=%d must be a multiple of 64 and >= 128 (zero-pad K)", kp);
    LLARK_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= kp && ldw >= kp && lda8 >= kp && lda8 % 16 == 0,
                  "gemm16_lo8: lda/ldw/lda8 must be >= kp, lda/ldw multiples of 8, lda8 a multiple of 16");
    LLARK_REQUIRE(((uintptr_t)a_hi & 15) == 0 && ((uintptr_t)wt & 15) == 0 && ((uintptr_t)a_lo8 & 15) == 0,
                  "gemm16_lo8: operands must be 16-byte aligned");
    LLARK_REQUIRE(sa >= 0 && sa <= 40 && sw >= -40 && sw <= 40, "gemm16_lo8: scale exponents out of range (sa=%d sw=%d)", sa, sw);
    LLARK_REQUIRE(ldw8 >= kp && ldw8 % 16 == 0 && ((uintptr_t)w8 & 15) == 0, "gemm16_lo8: w8 plane needs ldw8 >= kp, ldw8 %% 16 == 0 and 16-byte alignment");
    if (epilogue == EPI_F32 || epilogue == EPI_RESID) LLARK_REQUIRE(c && ldc >= n, "gemm16_lo8: fp32 output missing");
    if (epilogue == EPI_RESID) LLARK_REQUIRE(resid && ldr >= n, "gemm16_lo8: residual missing");
    if (epilogue == EPI_QGELU_SPLIT8)
        LLARK_REQUIRE(out_hi && out_lo8 && ldo >= n && ldo8 >= ((n + 63) & ~63) && ldo8 % 64 == 0,
                      "gemm16_lo8: split outputs missing (ldo8 must cover n rounded up to 64 and be a multiple of 64)");
    LLARK_REQUIRE(epilogue == EPI_F32 || epilogue == EPI_RESID || epilogue == EPI_QGELU_SPLIT8, "gemm16_lo8: unsupported epilogue %d", epilogue);
    LLARK_REQUIRE(ws->cus > 0 && ws->cus % 8 == 0, "gemm16_lo8: device with %d CUs is not supported (needs a multiple of 8)", ws->cus);
    GemmParams p = {};
    p.Ahi = a_hi; p.Alo = a_lo8; p.lda = lda; p.lda8 = lda8; p.Wt = wt; p.ldw = ldw; p.bias = bias;
    p.M = m; p.N = n; p.Kp = kp; p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr;
    p.Ohi = out_hi; p.Olo = out_lo8; p.ldo = ldo; p.ldo8 = ldo8; p.lo8_sa = sa; p.lo8_sw = sw; p.W8 = w8; p.ldw8 = ldw8;
    hipStream_t s = (hipStream_t)stream;
    if (ws_begin(ws, p, s)) { set_error("gemm16_lo8: workspace unusable"); return LLARK_ERR_LAUNCH; }
    const int rc = launch_gemm256_lo8n(p, epilogue, s, ws->cus);
    if (rc == -1000) { set_error("gemm16_lo8: problem outside the kernel's range (32-bit operand offsets)"); return LLARK_ERR_UNSUPPORTED; }
    ws_end(ws, cdiv(m, 256) * cdiv(n, 256), ws->cus / 8);
    return rc;
}

// acc + bias -> (exact GELU) -> 16-bit planes, optionally with a second copy of the hi plane: the producer side of a
// K-concatenated [hi | lo | hi] operand (A.W for fp32-class W = W_hi + W_lo as ONE non-split product against
// [W_hi | W_hi | W_lo]: the HTSAT linears of csrc/clap.hip's callers).
extern "C" int llark_gemm16_act(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                                const void* wt, int ldw, const float* bias, int m, int n, int kp, void* out_hi, void* out_lo,
                                void* out_hi_dup, int ldo, int act, llark_stream_t stream) {
    LLARK_REQUIRE(epilogue == EPI_SPLIT16 || epilogue == EPI_OUT16, "gemm16_act: epilogue must be SPLIT16 or OUT16");
    LLARK_REQUIRE(act == 0 || act == 2, "gemm16_act: act must be 0 (none) or 2 (exact GELU)");
    LLARK_REQUIRE(!out_hi_dup || epilogue == EPI_SPLIT16, "gemm16_act: the duplicate hi plane belongs to the SPLIT16 epilogue");
    return gemm16_impl(variant, dtype, split, epilogue, a_hi, a_lo, lda, wt, ldw, bias, m, n, kp, nullptr, 0, nullptr, 0, out_hi,
                       out_lo, ldo, 0, 0, 0, 0, 0, 0, stream, out_hi_dup, act);
}

extern "C" int llark_pack_weight16_frag(const void* wt, int ldw, int n, int kp, void* dst, llark_stream_t stream) {
    LLARK_REQUIRE(wt && dst && n > 0 && kp > 0 && kp % 64 == 0 && ldw >= kp && ldw % 8 == 0, "pack_weight16_frag: bad arguments (kp must be a multiple of 64)");
    LLARK_REQUIRE(((uintptr_t)wt & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pack_weight16_frag: pointers must be 16-byte aligned");
    const long long total = (long long)cdiv(n, 32) * (kp / 16) * 64;
    pack_frag_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>((const unsigned short*)wt, ldw, n, kp, (uint4*)dst, total);
    return check_launch("pack_weight16_frag");
}

// CUs of the current device (cached per device ordinal: a property of the hardware, not state of the library).
static int llark_device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
        cached[dev] = cus;
    }
    return cached[dev];
}

// Library choice (variant -1) of llark_gemm16_fragw_sk: is the product cut along K?
// plain 16-bit operands at < 1 round: the two-way K cut pays only on a deep K (down_proj, K = 11008: 903 -> 974 TFLOP/s; o_proj,
// K = 4096: 866 -> 756 against the 128x128 tiles) -- profiles/r02_streamk.txt
static bool fragw_sk_rule(bool split, int m, int n, int kp) {
    const long long tiles_bd0 = (long long)cdiv(m, CfgBD0::BM) * cdiv(n, CfgBD0::BN);
    return (split && tiles_bd0 < 2LL * llark_device_cus()) ||
           (!split && (4 * tiles_bd0 <= 2LL * llark_device_cus() || (tiles_bd0 < 2LL * llark_device_cus() && kp >= 8192)));
}
// plain operands, narrow output: the 128x128 tiles (3 workgroups per CU) -- except for bf16 once the DMA loop takes plain operands
// (GEMM_BDA >= 2): its 128x256 whole tiles beat them (o_proj at M = 2968: 104 vs 111 us, profiles/r05_gemm_bda_plain_ab.txt)
static bool fragw_small_tiles(int dtype, bool split, int epilogue, int n) {
    if (GEMM_BDA >= 2 && dtype == LLARK_BF16) return false;
    return !split && !IS_SWIGLU(epilogue) && n <= 4096;
}

// 1 when the library choice of llark_gemm16_fragw_sk runs this product as WHOLE 128x256 tiles of the per-tile kernel (no K cut, not the
// 128x128 tiles): the case in which llark_gemm16_fragw_rope_qkv -- which always runs whole 128x256 tiles -- is bit-equal to the two-launch
// path.  The one place this rule lives (the Python layer used to restate it).  Conservative: a product the K-cutting
// kernels would decline at launch time is reported as 0.
extern "C" int llark_gemm16_fragw_whole_tiles(int split, int epilogue, int m, int n, int kp) {
    if (m <= 0 || n <= 0 || kp <= 0) return 0;
    return (!fragw_sk_rule(split != 0, m, n, kp) && !fragw_small_tiles(LLARK_BF16, split != 0, epilogue, n)) ? 1 : 0;      // (asked by the bf16 Llama engine)
}

static int gemm16_fragw_impl(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                             const void* wfrag, const float* bias, int m, int n, int kp, float* c, int ldc,
                             const float* resid, int ldr, void* out_hi, void* out_lo, int ldo, void* scratch, long long scratch_bytes,
                             llark_stream_t stream) {
    LLARK_REQUIRE(a_hi && wfrag && m > 0 && n > 0 && kp > 0 && kp % 64 == 0, "gemm16_fragw: null pointer / empty problem / kp not a multiple of 64");
    LLARK_REQUIRE(lda % 8 == 0 && lda >= kp, "gemm16_fragw: lda must be >= kp and a multiple of 8");
    LLARK_REQUIRE(!split || a_lo, "gemm16_fragw: split mode needs the lo plane");
    LLARK_REQUIRE(((uintptr_t)a_hi & 15) == 0 && ((uintptr_t)wfrag & 15) == 0 && (!a_lo || ((uintptr_t)a_lo & 15) == 0),
                  "gemm16_fragw: operands must be 16-byte aligned");
    if (epilogue == EPI_F32 || epilogue == EPI_RESID) LLARK_REQUIRE(c && ldc >= n, "gemm16_fragw: fp32 output missing");
    if (epilogue == EPI_RESID) LLARK_REQUIRE(resid && ldr >= n, "gemm16_fragw: residual missing");
    if (epilogue == EPI_QGELU_SPLIT || epilogue == EPI_SPLIT16) LLARK_REQUIRE(out_hi && out_lo && ldo >= n, "gemm16_fragw: split outputs missing");
    if (epilogue == EPI_OUT16) LLARK_REQUIRE(out_hi && ldo >= n, "gemm16_fragw: 16-bit output missing");
    if (IS_SWIGLU(epilogue)) LLARK_REQUIRE(out_hi && n % 64 == 0 && ldo >= n / 2 && (epilogue != EPI_SWIGLU_SPLIT || out_lo), "gemm16_fragw: SwiGLU needs n%%64==0 and 16-bit outputs");
    GemmParams p = {};
    p.Ahi = a_hi; p.Alo = a_lo; p.lda = lda; p.Wt = wfrag; p.ldw = 0; p.bias = bias; p.M = m; p.N = n; p.Kp = kp;
    p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr; p.Ohi = out_hi; p.Olo = out_lo; p.ldo = ldo;
    hipStream_t s = (hipStream_t)stream;
    if (variant == 2) {                                             // gemm_bda.hip: hi + lo bf16 operands, A by LDS-DMA (same results as variant 0)
        const int rc = launch_gemm_bda(p, dtype, epilogue, s);
        if (rc == -1000) { set_error("gemm16_fragw: variant 2 takes bf16 operands with kp >= 192 and its epilogues only"); return LLARK_ERR_UNSUPPORTED; }
        return rc;
    }
    if (variant > 2) { set_error("gemm16_fragw: unknown variant %d", variant); return LLARK_ERR_INVALID; }
    // variant 1 (128x128 tiles, 3 workgroups per CU) fills the chip where the big tile leaves a ragged wave.  Measured
    // at M = 2968, bf16 (profiles/r01_gemm_variants.txt): it wins only on the narrow outputs -- o_proj 844 vs 798,
    // down_proj 939 vs 903 TFLOP/s (N = 4096: 768 small tiles = exactly one wave) -- and loses on qkv / lm_head
    // (822 vs 865, 852 vs 974), where bytes per flop matter more than wave quantisation.
    // Stream-K over the resident workgroups (128x256 tiles).  Measured at M = 2968 (profiles/r02_streamk.txt): it pays where
    // the whole product is less than one round of split-mode tiles (o_proj / down_proj: 384 tiles over 512 workgroups, +7 % /
    // +6 %); with one or more whole rounds ahead of the remainder the hardware's dynamic dispatch of one workgroup per tile
    // already hides the ragged last round (qkv, gate/up: +-1 %; lm_head: -6 %), and plain 16-bit operands at N <= 4096 are
    // better served by the 128x128 tiles below.  variant -1 applies that rule, variant 0 + scratch cuts whenever it can.
    const bool sk_rule = fragw_sk_rule(split != 0, m, n, kp);
    if (scratch && (variant == 0 || (variant < 0 && sk_rule)) && epilogue != EPI_QGELU_SPLIT) {
        int rc = -1000;                                             // -1000 = whole rounds / too small to cut -> the per-tile kernels below
        const bool uniform = variant < 0;                          // library choice: the position-independent cut
        rc = gemm_bd_sk_dispatch(dtype, p, split != 0, epilogue, s, scratch, scratch_bytes, uniform);
        if (rc != -1000) return rc;
    }
    if (variant < 0) variant = fragw_small_tiles(dtype, split != 0, epilogue, n) ? 1 : 0;
    if (variant == 1) {
        if (IS_SWIGLU(epilogue)) { set_error("gemm16_fragw: variant 1 (128x128 tiles) has no SwiGLU epilogue"); return LLARK_ERR_UNSUPPORTED; }
        if (dtype == LLARK_F16) return dispatch_bd<half_t, CfgBD1>(p, split != 0, epilogue, s);
        if (dtype == LLARK_BF16) return dispatch_bd<bf16_t, CfgBD1>(p, split != 0, epilogue, s);
    }
    if (dtype == LLARK_F16) return dispatch_bd<half_t, CfgBD0>(p, split != 0, epilogue, s);
    if (dtype == LLARK_BF16) return dispatch_bd<bf16_t, CfgBD0>(p, split != 0, epilogue, s);
    set_error("gemm16_fragw: unknown dtype %d", dtype);
    return LLARK_ERR_INVALID;
}

extern "C" int llark_gemm16_fragw(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                                  const void* wfrag, const float* bias, int m, int n, int kp, float* c, int ldc,
                                  const float* resid, int ldr, void* out_hi, void* out_lo, int ldo, llark_stream_t stream) {
    return gemm16_fragw_impl(variant, dtype, split, epilogue, a_hi, a_lo, lda, wfrag, bias, m, n, kp, c, ldc, resid, ldr, out_hi, out_lo,
                             ldo, nullptr, 0, stream);
}

// Bytes of scratch llark_gemm16_fragw_sk needs on the current device: one 128 KiB fp32 slab per resident workgroup + the fixed 64 KiB
// flag region at its end.
extern "C" long long llark_gemm16_sk_scratch_bytes(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        set_error("gemm16_sk_scratch_bytes: cannot query the current device");
        return -1;
    }
    return (long long)cus * 2 * (CfgBD0::BM * CfgBD0::BN * 4) + SK_FLAG_BYTES;
}

// llark_gemm16_fragw with the stream-K decomposition (gemm_bd_sk_kernel) whenever the tile count is not a whole number of
// rounds of the resident workgroups.  `scratch`: caller-owned device memory of >= llark_gemm16_sk_scratch_bytes() bytes,
// ZEROED once after allocation; launches that share it must be ordered on one stream.  variant: -1 / 0 = stream-K where it
// applies, 1 = the plain 128x128 tiles.  Same results as llark_gemm16_fragw up to the fp32 summation order over K.
extern "C" int llark_gemm16_fragw_sk(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                                     const void* wfrag, const float* bias, int m, int n, int kp, float* c, int ldc,
                                     const float* resid, int ldr, void* out_hi, void* out_lo, int ldo, void* scratch,
                                     long long scratch_bytes, llark_stream_t stream) {
    LLARK_REQUIRE(!scratch || ((uintptr_t)scratch & 15) == 0, "gemm16_fragw_sk: scratch must be 16-byte aligned");
    return gemm16_fragw_impl(variant, dtype, split, epilogue, a_hi, a_lo, lda, wfrag, bias, m, n, kp, c, ldc, resid, ldr, out_hi, out_lo,
                             ldo, scratch, scratch_bytes, stream);
}

// The Llama q|k|v product of a prefill with RoPE, the head split, the K-cache append and the transposed V-cache append in its
// epilogue (gemm_epilogue_rope_qkv): one launch instead of llark_gemm16_fragw (fp32 qkv) + llark_rope_split_heads.  bf16 only.
// a_hi / a_lo [batch * s][lda]: RMSNorm output planes (a_lo NULL = plain bf16 mode; then the three *_lo outputs must be NULL too).
// wfrag: llark_pack_weight16_frag of the [3 * nh * 128][kp] q|k|v weight whose q and k rows are permuted per head to
// [0..31 | 64..95 | 32..63 | 96..127] (v rows in natural order).  Always whole 128x256 tiles (no K cut): callers that want results
// bit-equal to the two-launch path use it where that path runs whole tiles too (m >= 2 rounds of tiles, see gemm16_fragw_impl).
static int gemm16_fragw_rope_qkv_impl(const void* a_hi, const void* a_lo, int lda, const void* wfrag, int kp, int batch, int s, int nh,
                                      int hd, int pos0, const float* cos_t, const float* sin_t, int max_pos, void* q, void* k_cache,
                                      void* vt_cache, void* q_lo, void* k_cache_lo, void* vt_cache_lo, int smax, void* v_rm, llark_stream_t stream) {
    LLARK_REQUIRE(a_hi && wfrag && cos_t && sin_t && q && k_cache && vt_cache, "gemm16_fragw_rope_qkv: null pointer");
    LLARK_REQUIRE(hd == 128, "gemm16_fragw_rope_qkv: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(nh > 0 && nh % 2 == 0, "gemm16_fragw_rope_qkv: a 256-column tile holds two heads: nh must be even, got %d", nh);
    LLARK_REQUIRE(batch > 0 && s > 0 && pos0 >= 0 && pos0 + s <= smax && pos0 + s <= max_pos && smax % 8 == 0,
                  "gemm16_fragw_rope_qkv: bad shape batch=%d s=%d pos0=%d smax=%d max_pos=%d", batch, s, pos0, smax, max_pos);
    LLARK_REQUIRE(kp > 0 && kp % 64 == 0 && lda % 8 == 0 && lda >= kp, "gemm16_fragw_rope_qkv: kp must be a multiple of 64, lda >= kp and a multiple of 8");
    LLARK_REQUIRE(s >= 32, "gemm16_fragw_rope_qkv: a 32-row block may cross one sequence boundary only: s must be >= 32, got %d", s);
    LLARK_REQUIRE((long long)batch * nh * smax * 256 < (1ll << 31) && (long long)max_pos * 256 < (1ll << 31) && (long long)batch * s < (1ll << 30),
                  "gemm16_fragw_rope_qkv: the planes are addressed with 32-bit byte offsets: batch * nh * smax * 256 must stay below 2 GiB");
    const bool split = a_lo != nullptr;
    LLARK_REQUIRE(split == (q_lo != nullptr) && split == (k_cache_lo != nullptr) && split == (vt_cache_lo != nullptr),
                  "gemm16_fragw_rope_qkv: give a_lo and all three lo outputs (fp32-class mode) or none");
    LLARK_REQUIRE(((uintptr_t)a_hi & 15) == 0 && ((uintptr_t)wfrag & 15) == 0 && (!a_lo || ((uintptr_t)a_lo & 15) == 0),
                  "gemm16_fragw_rope_qkv: operands must be 16-byte aligned");
    GemmParams p = {};
    p.Ahi = a_hi; p.Alo = a_lo; p.lda = lda; p.Wt = wfrag; p.M = batch * s; p.N = 3 * nh * 128; p.Kp = kp;
    p.rope_cos = cos_t; p.rope_sin = sin_t; p.rope_s = s; p.rope_nh = nh; p.rope_pos0 = pos0; p.rope_smax = smax;
    p.rope_q = q; p.rope_q_lo = q_lo; p.rope_k = k_cache; p.rope_k_lo = k_cache_lo; p.rope_v = vt_cache; p.rope_v_lo = vt_cache_lo;
    LLARK_REQUIRE(!v_rm || (!split && (long long)batch * nh * s * 256 < (1ll << 31)), "gemm16_fragw_rope_qkv: the row-major V output exists in the plain bf16 mode only");
    p.rope_v_rm = v_rm;
    hipStream_t st = (hipStream_t)stream;
    if (GEMM_BDA && (split || GEMM_BDA >= 2)) {
        const int rc = launch_gemm_bda(p, LLARK_BF16, EPI_ROPE_QKV, st);
        if (rc != -1000) return rc;
    }
    return split ? launch_gemm_bd<bf16_t, true, EPI_ROPE_QKV, CfgBD0>(p, st) : launch_gemm_bd<bf16_t, false, EPI_ROPE_QKV, CfgBD0>(p, st);
}

extern "C" int llark_gemm16_fragw_rope_qkv(const void* a_hi, const void* a_lo, int lda, const void* wfrag, int kp, int batch, int s, int nh,
                                           int hd, int pos0, const float* cos_t, const float* sin_t, int max_pos, void* q, void* k_cache,
                                           void* vt_cache, void* q_lo, void* k_cache_lo, void* vt_cache_lo, int smax, llark_stream_t stream) {
    return gemm16_fragw_rope_qkv_impl(a_hi, a_lo, lda, wfrag, kp, batch, s, nh, hd, pos0, cos_t, sin_t, max_pos, q, k_cache, vt_cache, q_lo,
                                      k_cache_lo, vt_cache_lo, smax, nullptr, stream);
}

// Training form (plain bf16 operands): llark_gemm16_fragw_rope_qkv that ALSO writes V row-major, v_rm [batch][nh][s][128] -- the operand
// the attention backward (llark_attn_backward_bf16*) reads, otherwise rebuilt from the V^T cache by a llark_transpose16 per layer and step.
extern "C" int llark_gemm16_fragw_rope_qkv_train(const void* a, int lda, const void* wfrag, int kp, int batch, int s, int nh, int hd, int pos0,
                                                 const float* cos_t, const float* sin_t, int max_pos, void* q, void* k_cache, void* vt_cache,
                                                 int smax, void* v_rm, llark_stream_t stream) {
    LLARK_REQUIRE(v_rm, "gemm16_fragw_rope_qkv_train: null v_rm");
    return gemm16_fragw_rope_qkv_impl(a, nullptr, lda, wfrag, kp, batch, s, nh, hd, pos0, cos_t, sin_t, max_pos, q, k_cache, vt_cache, nullptr,
                                      nullptr, nullptr, smax, v_rm, stream);
}

// Decode-step form of `h += x . W^T` followed by RMSNorm(h) -> bf16 planes, in ONE launch (m <= 16 rows): the skinny
// weight-streaming kernel plus a last-workgroup-done tail.  Replaces o_proj / down_proj + the following
// LlamaRMSNorm of m2t/models/llamav2.py:224-234 (HF LlamaDecoderLayer) in the cached decode path.
extern "C" int llark_gemm16_resid_rmsnorm(int dtype, int split, const void* a_hi, const void* a_lo, int lda, const void* wt,
                                          int ldw, int m, int n, int kp, float* h, int ldh, const float* norm_w, float eps,
                                          void* x_hi, void* x_lo, int ldx, llark_stream_t stream) {
    LLARK_REQUIRE(a_hi && wt && h && norm_w && x_hi && m > 0 && n > 0 && kp > 0 && kp % 32 == 0, "gemm16_resid_rmsnorm: bad arguments");
    LLARK_REQUIRE(m <= 16, "gemm16_resid_rmsnorm: decode form only (m <= 16), got m=%d", m);
    LLARK_REQUIRE(n % 4 == 0 && ldh % 4 == 0 && ldx % 4 == 0 && ldh >= n && ldx >= n && lda >= kp && ldw >= kp, "gemm16_resid_rmsnorm: bad leading dimensions");
    LLARK_REQUIRE(!split || (a_lo && x_lo), "gemm16_resid_rmsnorm: split mode needs the lo planes");
    LLARK_REQUIRE(dtype == LLARK_BF16, "gemm16_resid_rmsnorm: bf16 only");
    static int* counters = nullptr;
    static unsigned next = 0;
    if (!counters) {
        if (hipMalloc((void**)&counters, 64 * sizeof(int)) != hipSuccess || hipMemset(counters, 0, 64 * sizeof(int)) != hipSuccess) {
            set_error("gemm16_resid_rmsnorm: cannot allocate the arrival counters");
            return LLARK_ERR_LAUNCH;
        }
    }
    GemmParams p = {};
    p.Ahi = a_hi; p.Alo = a_lo; p.lda = lda; p.Wt = wt; p.ldw = ldw; p.M = m; p.N = n; p.Kp = kp;
    p.C = h; p.ldc = ldh; p.R = h; p.ldr = ldh;
    p.nw = norm_w; p.neps = eps; p.nhi = x_hi; p.nlo = x_lo; p.ldn = ldx; p.ncnt = counters + (next++ % 64);
    const int rc = dispatch_skinny<bf16_t>(p, split != 0, EPI_RESID, (hipStream_t)stream);
    if (rc == 1) { set_error("gemm16_resid_rmsnorm: no skinny kernel for this epilogue"); return LLARK_ERR_UNSUPPORTED; }
    return rc;
}

// Decode-step form of  RMSNorm(x) . W^T  in ONE launch (m <= 16 rows): the skinny weight-streaming kernel normalises its A
// rows on the fly (bit-identical to llark_rmsnorm_bf16 + llark_gemm16).  Replaces LlamaRMSNorm + {q,k,v}_proj / gate,up /
// lm_head of the cached decode path (m2t/models/llamav2.py:224-234,312 -> HF LlamaDecoderLayer).  epilogue: F32 or SwiGLU.
extern "C" int llark_gemm16_rmsnorm_a(int dtype, int split, int epilogue, const float* x, int ldx, const float* norm_w, float eps,
                                      const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c, int ldc,
                                      void* out_hi, void* out_lo, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(x && norm_w && wt && m > 0 && n > 0 && kp > 0 && kp % 32 == 0, "gemm16_rmsnorm_a: bad arguments");
    LLARK_REQUIRE(m <= 16, "gemm16_rmsnorm_a: decode form only (m <= 16), got m=%d", m);
    LLARK_REQUIRE(dtype == LLARK_BF16, "gemm16_rmsnorm_a: bf16 only");
    LLARK_REQUIRE(ldx % 4 == 0 && ldx >= kp && ldw >= kp && ((uintptr_t)x & 15) == 0 && ((uintptr_t)norm_w & 15) == 0,
                  "gemm16_rmsnorm_a: x / norm_w must be 16-byte aligned with ldx >= kp");
    LLARK_REQUIRE(epilogue == EPI_F32 || IS_SWIGLU(epilogue), "gemm16_rmsnorm_a: epilogue must be F32 or SwiGLU");
    if (epilogue == EPI_F32) LLARK_REQUIRE(c && ldc >= n, "gemm16_rmsnorm_a: fp32 output missing");
    if (IS_SWIGLU(epilogue)) LLARK_REQUIRE(out_hi && n % 64 == 0 && ldo >= n / 2 && (epilogue != EPI_SWIGLU_SPLIT || out_lo), "gemm16_rmsnorm_a: SwiGLU outputs missing");
    GemmParams p = {};
    p.Wt = wt; p.ldw = ldw; p.bias = bias; p.M = m; p.N = n; p.Kp = kp; p.C = c; p.ldc = ldc; p.Ohi = out_hi; p.Olo = out_lo; p.ldo = ldo;
    p.xn = x; p.xg = norm_w; p.ldxn = ldx; p.xeps = eps;
    const int rc = dispatch_skinny<bf16_t>(p, split != 0, epilogue, (hipStream_t)stream);
    if (rc == 1) { set_error("gemm16_rmsnorm_a: no skinny kernel for this epilogue"); return LLARK_ERR_UNSUPPORTED; }
    return rc;
}

extern "C" int llark_gemm16_batched(int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                                    long long stride_a, const void* wt, int ldw, long long stride_w, int m, int n, int kp,
                                    float* c, int ldc, long long stride_c, void* out_hi, void* out_lo, int ldo,
                                    long long stride_o, int batch, llark_stream_t stream) {
    LLARK_REQUIRE(batch >= 1 && batch <= 65535, "gemm16_batched: batch %d out of range", batch);
    return gemm16_impl(-1, dtype, split, epilogue, a_hi, a_lo, lda, wt, ldw, nullptr, m, n, kp, c, ldc, nullptr, 0, out_hi, out_lo,
                       ldo, batch, stride_a, stride_w, stride_c, 0, stride_o, stream);
}

extern "C" int llark_gemm16(int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                            const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c, int ldc,
                            const float* resid, int ldr, void* out_hi, void* out_lo, int ldo, llark_stream_t stream) {
    return llark_gemm16_ex(-1, dtype, split, epilogue, a_hi, a_lo, lda, wt, ldw, bias, m, n, kp, c, ldc, resid, ldr, out_hi,
                           out_lo, ldo, stream);
}
