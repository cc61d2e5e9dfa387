// Developer probe (NOT product code): the producer / consumer-wave variant of gram_cached_kernel<float> that VERDICT r5
// next #6 asked for, measured and closed in round 6 (profiles/r06_gram_pc_ab.md).  Included by tools/gram_probe.hip behind
// mvf_gram.hip, inside namespace mvf.
// ----------------------------------------------------------------------------------------------------------------
// float32 cached Gram with PRODUCER and CONSUMER waves (round 6; developer option "gram_pc").
// gram_cached_kernel<float> spends 10 v_cvt_f64_f32 + 2 v_mul_f64 per 16 MFMAs in EVERY wave, and inside one wave that
// VALU time is additive to the f64 MFMA time.  Here a 512-thread workgroup owns one 128 x 128 tile: waves 4 - 7 (one per
// SIMD) load the 16 operand blocks of a k-step from the cache ONCE per workgroup, widen them, scale the row blocks by P
// (exact in float64) and drop them as float64 into an LDS ring; waves 0 - 3 (one per SIMD) only ds_read_b64 + MFMA.  No
// workgroup barrier in the loop: per ring slot a `ready` counter (producers add 1 when their blocks are written) and a
// `consumed` counter (consumers add 1 when their reads of the slot have returned); LDS executes one wave's operations in
// order, so data-then-flag needs no fence.  The LDS traffic and the MFMAs are inline assembly in program order (the
// compiler merges ds_read_b64 pairs into ds_read2st64_b64, 4 x the LDS cycles, and sinks the prefetch below the MFMAs).
// Same values, same k-step order per output element: bit-identical to gram_cached_kernel<float>.
// Ring: PC_R slots of PC_KS k-steps x 16 blocks x 512 B; byte offset = slot * PC_SLOT + q * 8192 + block * 512 + lane * 8.
// ----------------------------------------------------------------------------------------------------------------
#ifndef MVF_PC_R
#define MVF_PC_R 8
#endif
#ifndef MVF_PC_PF
#define MVF_PC_PF 4  // producer register buffers (loads PF - 1 stages ahead of the stage being written)
#endif
#ifndef MVF_PC_PROBE
#define MVF_PC_PROBE 0  // 1: no producers, flags ignored; 2: also no LDS reads; 3: producers without global loads
#endif
constexpr int PC_KS = 2, PC_R = MVF_PC_R, PC_PF = MVF_PC_PF;
constexpr int PC_SLOT = PC_KS * 16 * 512;                 // 16 KiB
constexpr int PC_FLAGS = PC_R * PC_SLOT;                  // ready[PC_R] then consumed[PC_R] behind the ring
constexpr int PC_LDS_BYTES = PC_FLAGS + 2 * PC_R * 4;
static_assert(PC_R % 2 == 0 && PC_R * PC_SLOT <= 131072, "ring addressing: two base registers 64 KiB apart");

template <int I> using IC = std::integral_constant<int, I>;
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<N, I + 1>(f);
    }
}
// ring byte offset -> (which base register, immediate)
constexpr int pc_hi(int off) { return off >= 65536 ? 1 : 0; }
constexpr int pc_imm(int off) { return off >= 65536 ? off - 65536 : off; }

template <int OFF>
__device__ __forceinline__ void pc_ds_read_b64(double& d, unsigned addr) {
#if MVF_PC_PROBE == 2
    asm volatile("" : "+v"(d));
#else
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
#endif
}
template <int OFF>
__device__ __forceinline__ void pc_ds_write_b64(unsigned addr, double d) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(d), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void pc_ds_read_b32(unsigned& d, unsigned addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// lane 0 only: *(flags + OFF) += 1 (no return value); every wave here is full, so exec is restored to all ones
template <int OFF>
__device__ __forceinline__ void pc_bump(unsigned addr, unsigned one) {
    asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1 offset:%2\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(one), "n"(OFF)
                 : "memory");
}
__device__ __forceinline__ void pc_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// consumer wave: NB live column blocks (8, or 4 for the edge shape); DW >= 0: balanced diagonal shape (as cached_block)
template <int NB, int DW>
__device__ __forceinline__ void pc_consume(int wave, int nstages, double* __restrict__ out) {
    constexpr int NA = 2;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    auto blk_live = [](int a, int b) constexpr { return DW >= 0 ? b >= (a == 0 ? DW : 7 - DW) : b < NB; };
    constexpr int C0 = DW >= 0 ? DW : 0;            // first live column block
    constexpr int NC = DW >= 0 ? 8 - DW : NB;       // live column blocks C0 .. C0 + NC - 1
    constexpr int NRD = PC_KS * (NA + NC);          // operand reads per stage
    constexpr int per_q = DW >= 0 ? 9 : NA * NB;    // live MFMAs per k-step
    constexpr int NLIVE = PC_KS * per_q;
    const int r0 = DW >= 0 ? DW : 2 * wave, r1 = DW >= 0 ? 7 - DW : 2 * wave + 1;
    // base addresses (bytes): [which 64 KiB half][row block 0 / row block 1 / column blocks]
    unsigned base[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        base[h][0] = (unsigned)(h * 65536 + r0 * 512 + lane * 8);
        base[h][1] = (unsigned)(h * 65536 + r1 * 512 + lane * 8);
        base[h][2] = (unsigned)(h * 65536 + 8 * 512 + lane * 8);
    }
    const unsigned fb_addr = (unsigned)PC_FLAGS, one = 1u;
    f64x4 acc[NA][8];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    double op[2][PC_KS][NA + 8];  // [register buffer][k-step][2 row operands, then the column operands by block]
#if MVF_PC_PROBE == 2
#pragma unroll
    for (int i = 0; i < 2 * PC_KS * (NA + 8); ++i) (&op[0][0][0])[i] = 0.5 + lane * 1e-3 + i * 0.01;
#endif
    // read number r of a stage: k-step r / (NA + NC); operand (r % (NA + NC)) = row 0, row 1, column C0 ...
    auto rd = [&](auto SLOT, auto BUF, auto RI) {
        constexpr int slot = decltype(SLOT)::value, buf = decltype(BUF)::value, r = decltype(RI)::value;
        constexpr int q = r / (NA + NC), o = r % (NA + NC);
        constexpr int off = slot * PC_SLOT + q * 8192;
        if constexpr (o < NA)
            pc_ds_read_b64<pc_imm(off)>(op[buf][q][o], base[pc_hi(off)][o]);
        else
            pc_ds_read_b64<pc_imm(off) + (C0 + o - NA) * 512>(op[buf][q][NA + C0 + o - NA], base[pc_hi(off)][2]);
    };
    const int niter = nstages / PC_R;  // slices are multiples of 256 cells: nstages is a multiple of 32
    __builtin_amdgcn_s_setprio(2);
    unsigned fl;
    // prologue: stage 0 must be there, its operands into buffer 0, the flag of stage 1 behind them
    do {
        pc_ds_read_b32<0>(fl, fb_addr);
        pc_wait_lds();
    } while (fl < 4u && !(MVF_PC_PROBE == 1 || MVF_PC_PROBE == 2));
    static_for<NRD>([&](auto RI) { rd(IC<0>{}, IC<0>{}, RI); });
    pc_ds_read_b32<4 * (1 % PC_R)>(fl, fb_addr);
    for (int it = 0; it < niter; ++it) {
        static_for<PC_R>([&](auto SLOT) {
            constexpr int slot = decltype(SLOT)::value, buf = slot & 1, nslot = (slot + 1) % PC_R, n2slot = (slot + 2) % PC_R;
            const bool last_stage = (it == niter - 1) && (slot == PC_R - 1);
            const unsigned need = (MVF_PC_PROBE == 1 || MVF_PC_PROBE == 2) ? 0u : 4u * (unsigned)(slot + 1 == PC_R ? it + 2 : it + 1);
            pc_wait_lds();  // operands of this stage (issued a stage ago) and the flag of the next one are in registers
            pc_bump<PC_R * 4 + slot * 4>(fb_addr, one);  // consumed[slot]: this wave is done with the slot
            if (!last_stage) {
                while (fl < need) {
                    __builtin_amdgcn_s_sleep(1);
                    pc_ds_read_b32<nslot * 4>(fl, fb_addr);
                    pc_wait_lds();
                }
            }
            // MFMAs of this stage; behind each of the first ones a read of the next stage, then the flag of the one after
            static_for<PC_KS * NA * 8>([&](auto MI) {
                constexpr int mi = decltype(MI)::value, q = mi / (NA * 8), a = (mi / 8) % NA, b = mi % 8;
                if constexpr (blk_live(a, b)) {
                    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0"
                                 : "+v"(acc[a][b])
                                 : "v"(op[buf][q][a]), "v"(op[buf][q][NA + b]));
                    // index of this MFMA among the live ones = number of live blocks before it
                    constexpr int before_a = DW >= 0 ? (a == 0 ? 0 : 8 - DW) : a * NB;
                    constexpr int live_idx = q * per_q + before_a + (b - (DW >= 0 ? (a == 0 ? DW : 7 - DW) : 0));
                    if constexpr (live_idx < NRD) {
                        if (!last_stage) rd(IC<nslot>{}, IC<buf ^ 1>{}, IC<live_idx>{});
                    } else if constexpr (live_idx == NRD) {
                        if (!last_stage) pc_ds_read_b32<n2slot * 4>(fl, fb_addr);
                    }
                }
            });
            if constexpr (NLIVE <= NRD) {  // fewer MFMAs than reads (diagonal shapes): the rest behind them
                if (!last_stage) {
                    static_for<NRD - NLIVE>([&](auto RI) { rd(IC<nslot>{}, IC<buf ^ 1>{}, IC<NLIVE + decltype(RI)::value>{}); });
                    pc_ds_read_b32<n2slot * 4>(fl, fb_addr);
                }
            }
        });
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> VALU / stores (the compiler does not see the asm MFMAs)
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (blk_live(a, b)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (a == 0 ? r0 : r1) * 16 + lk + 4 * r;
                    const int col = b * 16 + li;
                    __builtin_nontemporal_store(acc[a][b][r], &out[row * GT + col]);
                }
            }
}

// producer wave p: row blocks 2p, 2p + 1 (scaled by P) and column blocks 2p, 2p + 1 (only if live)
__device__ __forceinline__ void pc_produce(int p, const float* __restrict__ ublk, const float* __restrict__ P, int64_t n,
                                           int64_t n_pad, int64_t n0, int nstages, int64_t rb, int64_t cb, bool need_b) {
    const int lane = threadIdx.x & 63, lk = lane >> 4;
    const float* ga[2];
    const float* gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        ga[j] = ublk + ((rb + 2 * p + j) * n_pad + n0) * UB + lane;
        gb[j] = ublk + ((cb + 2 * p + j) * n_pad + n0) * UB + lane;
    }
    const float* gp = P + n0 + lk;
    const int pmax = (int)min((int64_t)0x3fffffff, n - 1 - n0 - lk);  // may be < 0: then P[n - 1] (cached rows are zero)
    float va[PC_PF][PC_KS][2], vb[PC_PF][PC_KS][2], vp[PC_PF][PC_KS];
    auto gl = [&](int t, int buf) {
#pragma unroll
        for (int q = 0; q < PC_KS; ++q) {
            const int64_t ks = (int64_t)t * PC_KS + q;
#pragma unroll
#if MVF_PC_PROBE >= 3
            for (int j = 0; j < 2; ++j) va[buf][q][j] = 0.5f + lane * 1e-3f + (float)ks * 1e-9f, vb[buf][q][j] = 0.25f + lane * 1e-3f;
            vp[buf][q] = 1.0f;
#else
            for (int j = 0; j < 2; ++j) va[buf][q][j] = ga[j][ks * (4 * UB)];
            if (need_b) {
#pragma unroll
                for (int j = 0; j < 2; ++j) vb[buf][q][j] = gb[j][ks * (4 * UB)];
            }
            vp[buf][q] = gp[min((int)(ks * 4), pmax)];
#endif
        }
    };
    static_assert(PC_R % PC_PF == 0, "register buffers are static inside a ring revolution");
    unsigned wbase[2];  // this wave's row blocks in either 64 KiB half; column blocks are 8 * 512 bytes further
    wbase[0] = (unsigned)(2 * p * 512 + lane * 8);
    wbase[1] = wbase[0] + 65536u;
    const unsigned fb_addr = (unsigned)PC_FLAGS, one = 1u;
    const int niter = nstages / PC_R;
#pragma unroll
    for (int s = 0; s < PC_PF - 1; ++s) gl(s, s);
    for (int it = 0; it < niter; ++it) {
        static_for<PC_R>([&](auto SLOT) {
            constexpr int slot = decltype(SLOT)::value, buf = slot % PC_PF;
            const int t = it * PC_R + slot;
            if (t + PC_PF - 1 < nstages) gl(t + PC_PF - 1, (slot + PC_PF - 1) % PC_PF);
            // widen + scale first (VALU work while the slot may still be in use), then wait for the slot
            double da[PC_KS][2], db[PC_KS][2];
#pragma unroll
            for (int q = 0; q < PC_KS; ++q) {
#if MVF_PC_PROBE == 4 || MVF_PC_PROBE == 5
                for (int j = 0; j < 2; ++j) da[q][j] = 0.5 + lane * 1e-3, db[q][j] = 0.25 + lane * 1e-3;  // loop invariant
#else
                const double pd = (double)vp[buf][q];
#pragma unroll
                for (int j = 0; j < 2; ++j) da[q][j] = (double)va[buf][q][j] * pd;  // exact in float64
                if (need_b) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) db[q][j] = (double)vb[buf][q][j];
                }
#endif
            }
            const unsigned need = 4u * (unsigned)it;
            unsigned c;
            pc_ds_read_b32<PC_R * 4 + slot * 4>(c, fb_addr);
            pc_wait_lds();
            while (c < need) {
                __builtin_amdgcn_s_sleep(2);
                pc_ds_read_b32<PC_R * 4 + slot * 4>(c, fb_addr);
                pc_wait_lds();
            }
#if MVF_PC_PROBE != 4
            static_for<PC_KS * 2>([&](auto QJ) {
                constexpr int q = decltype(QJ)::value / 2, j = decltype(QJ)::value % 2;
                constexpr int off = slot * PC_SLOT + q * 8192 + j * 512;
                pc_ds_write_b64<pc_imm(off)>(wbase[pc_hi(off)], da[q][j]);
                if (need_b) pc_ds_write_b64<pc_imm(off) + 8 * 512>(wbase[pc_hi(off)], db[q][j]);
            });
#endif
            pc_bump<slot * 4>(fb_addr, one);  // ready[slot]: behind the data in this wave's LDS order
        });
    }
}

__global__ __launch_bounds__(512, 1) void gram_pc_kernel(const float* __restrict__ ublk, const float* __restrict__ P, int64_t n,
                                                         int64_t n_pad, int64_t m, int nt, int npairs, int64_t slice_len,
                                                         int64_t slice0, double* __restrict__ partial) {
    extern __shared__ __align__(16) unsigned char pc_lds[];  // PC_LDS_BYTES, based at LDS address 0 (no static LDS here)
    const int pair = blockIdx.x % npairs;
    const int64_t slice = blockIdx.x / npairs;
    int ti, tj;
    decode_pair(pair, nt, ti, tj);
    const int64_t n0 = (slice0 + slice) * slice_len;
    const int64_t n1 = min(n_pad, n0 + slice_len);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* out = partial + ((size_t)slice * npairs + pair) * (size_t)(GT * GT);
    constexpr int TB = GT / UB;
    const int64_t rb = (int64_t)ti * TB, cb = (int64_t)tj * TB;
    const int nstages = (int)((n1 - n0) / (4 * PC_KS));
    if (threadIdx.x < 2 * PC_R) reinterpret_cast<unsigned*>(pc_lds + PC_FLAGS)[threadIdx.x] = 0u;
    __syncthreads();
    const bool edge = ti != tj && m - (int64_t)tj * GT <= 4 * UB;
    if (wave >= 4) {
        const int p = wave - 4;
        if (MVF_PC_PROBE == 1 || MVF_PC_PROBE == 2) return;
        pc_produce(p, ublk, P, n, n_pad, n0, nstages, rb, cb, !(edge && p >= 2));
        return;
    }
    if (ti != tj) {
        if (edge)
            pc_consume<4, -1>(wave, nstages, out);
        else
            pc_consume<8, -1>(wave, nstages, out);
    } else {
        switch (wave) {
            case 0: pc_consume<8, 0>(wave, nstages, out); break;
            case 1: pc_consume<8, 1>(wave, nstages, out); break;
            case 2: pc_consume<8, 2>(wave, nstages, out); break;
            default: pc_consume<8, 3>(wave, nstages, out); break;
        }
    }
}

