// The B-direct product cut along K: stream-K runs over the resident workgroups, or every tile cut into the same K ranges (see the header
// comment of gemm_bd_sk_kernel).  Llama-2 o_proj / down_proj at prefill sizes (m2t/models/llamav2.py:224-234 -> HF LlamaDecoderLayer) are
// less than one round of 128x256 tiles: llark_gemm16_fragw_sk routes them here.  Its own translation unit since round 6: the
// bf16 instantiations inline gemm_bda_loop.h's hand-counted DMA loop, whose generated code tests/test_gemm_bda_isa_cpu.py audits.
#include "gemm_bd.h"
#include "gemm_bda_loop.h"

namespace llark {

// ------------------------------------------------------------------------------------------
// Stream-K form of the B-direct kernel.  The Llama prefill products have M = 2968 rows = 24 row tiles, so the tile count
// is never a multiple of the 512 resident workgroups: 384 tiles (o_proj / down_proj) leave a quarter of the chip idle,
// 1152 (qkv) and 2064 (gate_up) pay a whole extra round for 128 and 16 left-over tiles.  Here the grid is exactly the
// resident set (S workgroups).  The last `sk_tiles` tiles of the linear tile order are cut into S equal runs of
// K-steps (`sk_per` each, a run may cross a tile boundary), every workgroup takes one run first and then its share of
// the remaining tiles whole ("data-parallel" rounds).  A tile whose K range is shared by several workgroups is
// finished by the one that holds its k = 0 end (the OWNER): the others (CONTRIBUTORS; for them the piece is always the
// first thing they do, so it is ready long before the owner needs it) write their fp32 accumulators to a slab in the
// caller's scratch with write-through (sc1) 16-byte stores, drain, and raise a flag; the owner polls the flags of the
// following slots in slot order, adds the slabs in that fixed order (deterministic: no atomics on data) and runs the
// ordinary fused epilogue.  Hand-off protocol: sc1 payload -> every wave s_waitcnt vmcnt(0) -> barrier -> one relaxed
// agent-scope flag store; consumer: one lane polls relaxed, one agent-scope acquire, barrier, plain loads.  The owner
// re-zeroes the flag, so a scratch that starts zeroed stays valid from launch to launch (launches sharing a scratch
// must be ordered on one stream).  All S workgroups must be resident (the host sizes the grid from the occupancy query).
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned sk_flag_t;
// The hand-off flags live in a FIXED region: the last SK_FLAG_BYTES of the caller's scratch, whatever the shape, tile configuration
// or decomposition of a launch (the fp32 slabs grow from the front and never reach it: launch_gemm_bd_sk checks).  "Zeroed once,
// every owner re-zeroes what it consumed" therefore holds across launches of different shapes sharing one scratch.

template <typename T, bool SPLIT, int EPI, typename C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm_bd_sk_kernel(const GemmParams p) {
    typedef typename Mfma<T>::frag frag;
    static_assert(C::WM == 1 && C::BK == 64, "B-direct layout: waves side by side over N, K-step 64");
    constexpr int ASTAGE = (SPLIT ? 2 : 1) * C::A_BYTES;
    constexpr int OFF_L = C::A_BYTES;
    constexpr int SLAB4 = C::BM * C::BN / 4;                             // float4 per slab
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 A stages

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = 0, wn = w;
    const int S = (int)gridDim.x;
    const int ks = p.sk_ks;
    // Block id -> slot (slot s works on units [s * sk_per, (s + 1) * sk_per) of the cut tiles).  Workgroups are dealt round-robin to
    // the 8 XCDs, so the map hands every XCD a contiguous band of runs / tiles (speed only).  Uniform split: the block order is
    // PIECE-major -- blocks [q * T, (q + 1) * T) hold piece q of every tile -- so the owner of a tile (its last piece) has a higher
    // block id than every piece it waits for WHATEVER the XCD remap does inside a piece plane: it only ever waits for workgroups
    // dispatched before it (scripts/sim_streamk_plan.py: slot_of_block).
    int slot;
    if (ks) {
        const int T_ = S / ks, piece = (int)blockIdx.x / T_, tb = (int)blockIdx.x - piece * T_;
        const int tile = (T_ & 7) ? tb : (tb & 7) * (T_ >> 3) + (tb >> 3);
        slot = tile * ks + piece;
    } else {
        slot = (S & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (S >> 3) + ((int)blockIdx.x >> 3);
    }
    // slab / flag of the contributor in slot s: stream-K runs: s itself (at most one shared piece per slot, its first);
    // uniform split (sk_ks pieces per tile, the LAST piece = owner): tile * (sk_ks - 1) + piece
    auto slab_of = [&](int s) { return ks ? (s / ks) * (ks - 1) + (s % ks) : s; };
    const int nk = p.Kp / C::BK;
    const int nk16 = nk * 4;
    const int rtiles = (p.N + 31) >> 5;
    const int sk_units = (p.tiles_m * p.tiles_n - p.sk_dp) * nk;
    int u0 = slot * p.sk_per;
    u0 = u0 < sk_units ? u0 : sk_units;
    const int u1 = (u0 + p.sk_per) < sk_units ? (u0 + p.sk_per) : sk_units;
    int dp_next = slot;

    const T* Ahi = (const T*)p.Ahi;
    const T* Alo = SPLIT ? (const T*)p.Alo : nullptr;
    constexpr int APW = (C::BM / C::RPI) / C::NW;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int a_rl = lane / C::CH, a_ch = lane % C::CH;
    float4* const part = (float4*)p.sk_part;
    sk_flag_t* const flags = (sk_flag_t*)p.sk_flag;

    for (;;) {
        int L, kt0, kt1;                                                  // next piece of work: tile L, K-steps [kt0, kt1)
        if (u0 < u1) {
            const int t = u0 / nk;
            L = p.sk_dp + t;
            kt0 = u0 - t * nk;
            kt1 = (kt0 + (u1 - u0)) < nk ? (kt0 + (u1 - u0)) : nk;
            u0 += kt1 - kt0;
        } else if (dp_next < p.sk_dp) {
            L = dp_next;
            dp_next += S;
            kt0 = 0;
            kt1 = nk;
        } else {
            break;
        }
        constexpr int GM = 8;
        const int gsz = GM * p.tiles_n;
        const int g = L / gsz;
        const int first_m = g * GM;
        const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
        const int tile_m = first_m + (L % gsz) % gm;
        const int tile_n = (L % gsz) / gm;
        const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;

        u32x4 areg[(SPLIT ? 2 : 1) * APW];
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
        const frag* wbase[C::TN];
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            int R = (n0 >> 5) + wn * C::TN + tn;
            R = R < rtiles ? R : rtiles - 1;
            wbase[tn] = (const frag*)p.Wt + ((size_t)R * nk16) * 64 + lane;
        }
        frag ring[4][C::TN];
        const int qlast = kt1 * 4 - 1;
        auto loadB = [&](int slot_, int q) {
            q = q < qlast ? q : qlast;                                     // tail: harmless re-load of the piece's last chunk
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) ring[slot_][tn] = wbase[tn][(size_t)q * 64];
        };

        f32x16_t acc[C::TM][C::TN];
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        // hi + lo bf16 pieces on the 128x256 tile run gemm_bda_loop.h's K loop (A by LDS-DMA, fragments read ahead): same arithmetic and
        // order per accumulator, so the slabs and the combined tiles are bit-identical to the register-staged loop below
        constexpr bool USE_BDA = GEMM_BDA && (SPLIT || GEMM_BDA >= 2) && std::is_same<T, bf16_t>::value && C::BM == 128 && C::BN == 256 && C::TM == 4 && C::TN == 2;
        if constexpr (USE_BDA) {
            bda_kloop<T, SPLIT>(p, smem, m0, n0, w, lane, kt0, kt1, acc);
        } else {
        loadA(kt0);
        loadB(0, kt0 * 4);
        loadB(1, kt0 * 4 + 1);
        loadB(2, kt0 * 4 + 2);
        storeA(0);
        __syncthreads();
        auto load_b1 = [&](int slot_, int tn, int q) __attribute__((always_inline)) {
            q = q < qlast ? q : qlast;                                     // tail: harmless re-load of the piece's last chunk
            ring[slot_][tn] = wbase[tn][(size_t)q * 64];
        };
        auto kloop = [&](auto ntm_tag) __attribute__((always_inline)) {
            constexpr int NTM = decltype(ntm_tag)::value;
            for (int kt = kt0; kt < kt1; ++kt) {
                loadA(kt + 1 < kt1 ? kt + 1 : kt);
                __builtin_amdgcn_sched_barrier(0);
                const char* sA = smem + ((kt - kt0) & 1) * ASTAGE;
                bd_kstep<T, SPLIT, C, NTM>(sA, sA + OFF_L, lane, ring, acc, kt * 4, load_b1);
                if (kt + 1 < kt1) storeA((kt + 1 - kt0) & 1);
                __syncthreads();
            }
        };
        if (GEMM_BD_TAIL_SKIP && C::TM > 1 && p.M - m0 <= 32) kloop(std::integral_constant<int, 1>{});  // ragged last row tile (see bd_kstep)
        else kloop(std::integral_constant<int, C::TM>{});
        }

        // Who finishes a shared tile.  Uniform split: the piece with the LAST K range -- the block order is piece-major (see `slot`
        // above), so it has a higher block id than the other pieces of its tile: it only ever waits for workgroups that were
        // dispatched before it and never wait themselves: no deadlock however few workgroup slots the device has free (other
        // streams, other processes).  Stream-K runs: the piece with the k = 0 end; the
        // pieces it waits for sit in the following slots, which is safe only because that form's grid is exactly the resident set
        // of an otherwise idle device (it is opt-in: variant 0).
        if (ks ? (kt1 < nk) : (kt0 != 0)) {
            // contributor: slab[slot] <- accumulators in register order (thread-contiguous 16-byte pieces), write-through
            // (buffer stores: one descriptor, one per-lane offset, the slab position as a scalar offset -- no address VGPRs)
            const int mine = slab_of(slot);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(part + (size_t)mine * SLAB4), 0, SLAB4 * 16, 0x00020000u);
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        // (whole-vector bit casts only: hipcc 7.2 miscompiles __builtin_bit_cast of a vector ELEMENT -- it reads element 0)
                        const f32x4_t f = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), rs, (int)threadIdx.x * 16, ((i * C::TN + j) * 4 + r4) * C::THREADS * 16, /*sc1: write-through*/ 16);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(flags + mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        int done = ks ? 0 : kt1;
        for (int cs = ks ? slot - (ks - 1) : slot + 1; ks ? (cs < slot) : (done < nk); ++cs) {       // owner: add the other pieces in slot order
            const int theirs = slab_of(cs);
            if (threadIdx.x == 0) {
                while (__hip_atomic_load(flags + theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(4);
                __hip_atomic_store(flags + theirs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(part + (size_t)theirs * SLAB4), 0, SLAB4 * 16, 0x00020000u);
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)threadIdx.x * 16, ((i * C::TN + j) * 4 + r4) * C::THREADS * 16, 0);
                        const f32x4_t v = __builtin_bit_cast(f32x4_t, u);
                        acc[i][j][4 * r4] += v[0];
                        acc[i][j][4 * r4 + 1] += v[1];
                        acc[i][j][4 * r4 + 2] += v[2];
                        acc[i][j][4 * r4 + 3] += v[3];
                    }
            done += (nk - done) < p.sk_per ? (nk - done) : p.sk_per;
        }
        gemm_epilogue<T, SPLIT, EPI, C>(p, acc, m0, n0, wm, wn, lane, 0);
    }
}

template <typename T, bool SPLIT, int EPI, typename C>
static int launch_gemm_bd_sk(GemmParams p, hipStream_t s, void* scratch, long long scratch_bytes, bool uniform) {
    constexpr int LDS = 2 * (SPLIT ? 2 : 1) * C::A_BYTES;
    auto kern = gemm_bd_sk_kernel<T, SPLIT, EPI, C>;
    static PerDeviceOnce once;                                     // resident workgroups per CU, per device
    if (once.first()) {
        int n = 0;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, C::THREADS, LDS) != hipSuccess) n = 0;
        once.slot() = n;
    }
    const int per_cu = once.slot();
    if (GEMM_BDA && (SPLIT || GEMM_BDA >= 2) && std::is_same<T, bf16_t>::value && C::BM == 128 && C::BN == 256 && ((long long)p.M * p.lda * 2 >= (1ll << 31) || p.Kp < 192))
        return -1000;                                              // the DMA loop addresses A with 32-bit byte offsets: the per-tile kernels take it
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1000;
    const int S = per_cu * cus;
    if (S <= 0 || S % 8) return -1000;
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    const int T_ = p.tiles_m * p.tiles_n, nk = p.Kp / C::BK;
    const long long slab_bytes = (long long)C::BM * C::BN * 4;
    if (uniform) {
        // Uniform split: EVERY tile is cut into the same `ks` K ranges, one workgroup each (grid = ks x tiles, dispatched
        // dynamically; the LAST piece of a tile finishes it, adding pieces 0 .. ks-2 in that order).  The summation tree of an output element is then the same wherever its
        // tile sits, so equal rows of a batch give bit-equal results (stream-K runs cut each tile at a position-dependent k).
        const int ks_ = (4 * T_ <= S && nk % 4 == 0 && nk / 4 >= 12) ? 4 : 2;
        if (T_ >= S || nk % ks_ || nk / ks_ < 12) return -1000;
        const long long need_u = (long long)T_ * (ks_ - 1) * slab_bytes + SK_FLAG_BYTES;
        if (!scratch || scratch_bytes < need_u || (long long)T_ * (ks_ - 1) * 4 > SK_FLAG_BYTES) return -1000;
        p.sk_ks = ks_;
        p.sk_dp = 0;
        p.sk_per = nk / ks_;
        p.sk_part = (float*)scratch;
        p.sk_flag = (unsigned*)((char*)scratch + scratch_bytes - SK_FLAG_BYTES);
        kern<<<dim3(T_ * ks_), C::THREADS, LDS, s>>>(p);
        return check_launch("gemm_bd_sk(uniform)");
    }
    p.sk_ks = 0;
    const int rounds = T_ / S, rem = T_ - rounds * S;
    if (rem == 0) return -1000;                                    // whole rounds: one workgroup per tile is already balanced
    const int sk_tiles = rem + (rounds >= 1 ? S : 0);               // with a full round in the pool no run is shorter than a tile
    const int per = cdiv(sk_tiles * nk, S);
    if (per < 12 || (long long)per * S < (long long)sk_tiles * nk) return -1000;                                    // pieces too short to pay for their prologue and the slab exchange
    const long long need = (long long)S * C::BM * C::BN * 4 + SK_FLAG_BYTES;
    if (!scratch || scratch_bytes < need || (long long)S * 4 > SK_FLAG_BYTES) return -1000;
    p.sk_dp = T_ - sk_tiles;
    p.sk_per = per;
    p.sk_part = (float*)scratch;
    p.sk_flag = (unsigned*)((char*)scratch + scratch_bytes - SK_FLAG_BYTES);
    kern<<<dim3(S), C::THREADS, LDS, s>>>(p);
    return check_launch("gemm_bd_sk");
}

template <typename T, typename C>
static int dispatch_bd_sk(const GemmParams& p, bool split, int epi, hipStream_t s, void* scratch, long long scratch_bytes, bool uniform) {
#define CASE(E)                                                      \
    case E:                                                          \
        return split ? launch_gemm_bd_sk<T, true, E, C>(p, s, scratch, scratch_bytes, uniform) : launch_gemm_bd_sk<T, false, E, C>(p, s, scratch, scratch_bytes, uniform);
    switch (epi) {
        CASE(EPI_F32)
        CASE(EPI_RESID)
        CASE(EPI_OUT16)
        CASE(EPI_SPLIT16)
        CASE(EPI_SWIGLU16)
        CASE(EPI_SWIGLU_SPLIT)
    }
#undef CASE
    return -1000;
}

int gemm_bd_sk_dispatch(int dtype, const GemmParams& p, bool split, int epi, hipStream_t s, void* scratch, long long scratch_bytes, bool uniform) {
    if (dtype == LLARK_F16) return dispatch_bd_sk<half_t, CfgBD0>(p, split, epi, s, scratch, scratch_bytes, uniform);
    if (dtype == LLARK_BF16) return dispatch_bd_sk<bf16_t, CfgBD0>(p, split, epi, s, scratch, scratch_bytes, uniform);
    return -1000;
}

}  // namespace llark
