// Object-per-lane form of the single-pass fit_predict kernel (bruteforce.py:602-631).
//
// k_fused gives a lane a MODEL and keeps the wave's few objects wave-uniform; every step
// then re-reads the model record and the object rows from LDS, and the per-object softmax
// state lives across lanes (wave reductions, a re-base every 64 steps, a block barrier per
// tile).  Here the roles are swapped: a lane owns OPL OBJECTS for the whole kernel (their
// fluxes / variances and running statistics sit in its VGPRs) and the models stream past as
// wave-uniform values -- scalar loads of the model records, SGPR operands to the fp64 ALU.
// No LDS traffic in the loop apart from the exp table, no cross-lane operation, no barrier.
//
//   k_ol      : one pass over the models; per object the weight-space running sum / max
//               (see fused_tile_w) and the candidate bits
//   k_ol_pdf  : per object: exact threshold, kernel stack, normalise (the PDF stage of k_fused)
//
// Candidates are kept as one BIT per (object, model) -- M/8 bytes per object, written as
// whole 256-B rows, no overflow case -- and the few per cent of pairs that are flagged have
// their likelihood recomputed in the PDF stage.
// Built for the mask-free weight-space likelihoods (SRC::WPOW == 3, dim_prior on).
//
// STATUS: alternative path, selected with FZ_OL=1 (parity-tested, not the default).  Measured on
// MI355X, 262144 x 1e5 x 5 (profiles/README.md): the model loop runs at 5.7e11 evals/s with
// band-constant model errors (46 ms; k_fused's loop: 61 ms) and 4.5e11 with per-model errors
// (58 ms; k_fused: 78 ms) -- but on the SURVEY 8d data 7 % of all pairs pass the weight threshold
// (7 242 of 7 255 flagged models per object are finally stacked), and handing that many pairs to
// the PDF stage through bits + recomputation costs 35-44 ms against 13 ms for k_fused's
// in-kernel candidate lists, so the totals are 82 vs 74 ms and 102 vs 92 ms.  The path wins
// only when posteriors are narrow (few pairs above the threshold).
#pragma once
#include "fz_kernels.h"

namespace fz {

struct OlStats { double ref, s, wmax; };      // per object: reference ln-like, sum and max of w = exp(lnl - ref)

// candidate bit of (object i, model j): word [i / 64][j / 32][i % 64], bit j % 32 -- a wave of
// k_ol (64 consecutive objects) writes each word row as one contiguous 256-B store
__device__ __forceinline__ size_t ol_word(int64_t group, int W, int w, int lane) { return ((size_t)group * W + w) * 64 + lane; }

template <class SRC, int OPL>
__global__ __launch_bounds__(256) void k_ol(SRC src_, int64_t N, int M, double wt_thresh, uint32_t* __restrict__ mask,
                                            OlStats* __restrict__ stats) {
    constexpr int WP = SRC::WPOW;
    static_assert(WP == 3, "object-per-lane kernel is built for the chi2^(3/2) likelihoods");
    SRC src = src_;
    const int tid = threadIdx.x, lane = tid & 63;
    // the exp table is read from global memory (vector L1, vmcnt): an LDS lookup would share
    // the lgkmcnt counter with the scalar model loads and make every wait on a table entry
    // also a wait on the prefetch of the next model record
    src.tb = global_tabs();
    const FastTabs tb = src.tb;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (tid >> 6);
    const int64_t obase = gw * (64 * OPL);
    if (obase >= N) return;
    const double thrf = (wt_thresh > 0.0) ? wt_thresh * 0.999000499833375 : 0.0;
    const double lg = src.lp.lg_full;
    const int W = (M + 31) >> 5;

    typename SRC::OR ob[OPL];
    int64_t oi[OPL];
    double ref[OPL], kref[OPL], s[OPL], wmax[OPL];
    uint32_t bits[OPL];
    typename SRC::MR m;
    src.load_model_rec(0, m);
#pragma unroll
    for (int o = 0; o < OPL; ++o) {
        oi[o] = obase + o * 64 + lane;
        src.load_obj(oi[o] < N ? oi[o] : N - 1, ob[o]);      // per-lane object: vector loads, once
        // reference = the first model's ln-like (chi2 == 0 -> -inf: fall back to 0)
        const double l0 = src.lnl_of_chi2(src.chi2_of(ob[o], m));
        ref[o] = (l0 - l0 == 0.0) ? l0 : 0.0;
        kref[o] = lg + ref[o];
        s[o] = 0.0; wmax[o] = 0.0; bits[o] = 0u;
    }
    for (int j = 0; j < M; ++j) {
        typename SRC::MR nx;
        src.load_model_rec(j + 1 < M ? j + 1 : j, nx);           // scalar prefetch of the next record
        double c2[OPL], t[OPL];
        bool over = false;
#pragma unroll
        for (int o = 0; o < OPL; ++o) {
            c2[o] = src.chi2_of(ob[o], m);
            t[o] = fma(-0.5, c2[o], -kref[o]);
            over |= t[o] > 500.0;
        }
        if (__any(over)) {                    // rare: a model e^500 better than the lane's reference
#pragma unroll
            for (int o = 0; o < OPL; ++o) {
                if (t[o] > 500.0) {
                    if (c2[o] > 0.0) {
                        const double nr = src.lnl_of_chi2(c2[o]);
                        const double f = exp_neg(ref[o] - nr, tb);
                        s[o] *= f; wmax[o] *= f;
                        ref[o] = nr; kref[o] = lg + nr;
                        t[o] = fma(-0.5, c2[o], -kref[o]);
                    } else {
                        t[o] = -700.0;             // chi2 == 0 (self match): lnl = -inf, w = 0 whatever the reference
                    }
                }
            }
        }
        const uint32_t bit = 1u << (j & 31);
#pragma unroll
        for (int o = 0; o < OPL; ++o) {
            const double e = exp_core(vmax_raw(t[o], -700.0), tb);
            const double cc = c2[o] + 1e-300;                    // chi2 == 0 (self match): w -> 0, no 0*inf
            const double y = __builtin_amdgcn_rsq(cc);
            double sq = cc * y;
            const double r = fma(-sq, 0.5 * y, 0.5);
            sq = fma(sq, r, sq);
            const double w = (c2[o] * sq) * e;
            s[o] += w;
            wmax[o] = vmax_raw(wmax[o], w);
            bits[o] |= (w > wmax[o] * thrf) ? bit : 0u;          // superset of wt > wt_thresh * max(wt): the max only grows
        }
        if ((j & 31) == 31 || j == M - 1) {
#pragma unroll
            for (int o = 0; o < OPL; ++o) { mask[ol_word(gw * OPL + o, W, j >> 5, lane)] = bits[o]; bits[o] = 0u; }
        }
        m = nx;
    }
#pragma unroll
    for (int o = 0; o < OPL; ++o) {
        if (oi[o] < N) { OlStats st; st.ref = ref[o]; st.s = s[o]; st.wmax = wmax[o]; stats[oi[o]] = st; }
    }
}

// PDF stage: a block of NWV waves takes NWV consecutive objects (one per wave).  The objects'
// candidate words are fetched by the whole block, FZ_OL_TILE word rows at a time, as 4*NWV-byte
// row segments and parked transposed in LDS; each wave then expands its own words, 4096 models
// at a time, into a list of model indices, recomputes those candidates' likelihoods 256 at a
// time (a lane per candidate), applies the exact threshold (pdf.py:510 / 591) against the final
// max and evidence, and stacks the kernels.
#define FZ_OL_TILE 256          // word rows per cooperative fetch
#define FZ_OL_SEG 4096          // models expanded per list (128 words: two per lane)
template <class SRC, int NWV>
__global__ __launch_bounds__(NWV * 64) void k_ol_pdf(SRC src_, const KdeView* __restrict__ kvp, int acc_stride, int64_t N, int M,
                                                     double wt_thresh, int normalize, const uint32_t* __restrict__ mask,
                                                     const OlStats* __restrict__ stats, double* __restrict__ lmap,
                                                     double* __restrict__ levid, double* __restrict__ pdfs) {
    static_assert(64 % NWV == 0, "a block's objects must share a 64-object group");
    extern __shared__ double smem[];     // [NWV][acc_stride] PDF rows | [NWV][FZ_OL_TILE] words | [NWV][FZ_OL_SEG] uint16
    SRC src = src_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    src.tb = global_tabs();
    const FastTabs tb = src.tb;
    const int64_t i0 = (int64_t)blockIdx.x * NWV;
    const int64_t i = i0 + wave;
    const bool live = i < N;
    double* row = smem + (size_t)wave * acc_stride;
    uint32_t* tile = reinterpret_cast<uint32_t*>(smem + (size_t)NWV * acc_stride);
    unsigned short* list = reinterpret_cast<unsigned short*>(tile + NWV * FZ_OL_TILE) + (size_t)wave * FZ_OL_SEG;
    const KdeView kv = *kvp;
    OlStats st; st.ref = 0.0; st.s = 1.0; st.wmax = 1.0;
    if (live) st = stats[i];
    const double le = st.ref + log_pos(st.s, tb);
    const double mx = st.ref + log_pos(st.wmax, tb);             // refined below from the candidates
    const bool ok = live && (le - le == 0.0);
    const int W = (M + 31) >> 5;
    const int64_t group = i0 >> 6; const int o0 = (int)(i0 & 63);
    typename SRC::OR ob;
    src.load_obj(live ? i : N - 1, ob);
    double lbest = -INFINITY;
    for (int k = lane; k < acc_stride; k += 64) row[k] = 0.0;
    const double thr = wt_thresh * exp_neg(mx - le, tb);
    for (int wt0 = 0; wt0 < W; wt0 += FZ_OL_TILE) {
        __syncthreads();                                          // the previous tile has been consumed
        for (int e = tid; e < FZ_OL_TILE * NWV; e += NWV * 64) {
            const int r = e / NWV, cidx = e % NWV;
            uint32_t word = 0u;
            if (wt0 + r < W && i0 + cidx < N) word = mask[ol_word(group, W, wt0 + r, o0 + cidx)];
            tile[cidx * FZ_OL_TILE + r] = word;
        }
        __syncthreads();
        if (!ok) continue;
        for (int sub = 0; sub < FZ_OL_TILE; sub += FZ_OL_SEG / 32) {
            if (wt0 + sub >= W) break;
            // lane l expands words 2l and 2l+1 of the segment (consecutive models stay consecutive in the list)
            uint32_t wa = tile[wave * FZ_OL_TILE + sub + 2 * lane], wb = tile[wave * FZ_OL_TILE + sub + 2 * lane + 1];
            if (__ballot((wa | wb) != 0u) == 0ull) continue;
            const int pc = __popc(wa) + __popc(wb);
            int inc = pc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
            const int total = __shfl(inc, 63, 64);
            int pos = inc - pc;
            while (wa) { const int b = __ffs((int)wa) - 1; list[pos++] = (unsigned short)(lane * 64 + b); wa &= wa - 1u; }
            while (wb) { const int b = __ffs((int)wb) - 1; list[pos++] = (unsigned short)(lane * 64 + 32 + b); wb &= wb - 1u; }
            const int jbase = (wt0 + sub) * 32;
            // four 64-candidate blocks in flight per trip (record gathers, then label gathers)
            constexpr int U = 4;
            for (int c0 = 0; c0 < total; c0 += 64 * U) {
                bool in[U], sel[U]; int jj[U]; double w[U];
                typename SRC::MR m[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = c0 + u * 64 + lane;
                    in[u] = k < total;
                    jj[u] = jbase + (int)list[in[u] ? k : 0];
                    src.load_model_rec16(jj[u], m[u]);             // per-lane gather of the 80-B record
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    double l = src.lnl_of_chi2(src.chi2_of(ob, m[u]));
                    l = in[u] ? l : -INFINITY;
                    lbest = fmax(lbest, l);
                    w[u] = exp_neg(l - le, tb);
                    sel[u] = in[u] && (w[u] > thr);
                }
                if (kv.kmode == KDE_HIST) {
                    int p[U]; double nr[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { const int j = sel[u] ? jj[u] : 0; p[u] = kv.pos[j]; nr[u] = kv.norm[j]; }
#pragma unroll
                    for (int u = 0; u < U; ++u) if (sel[u]) unsafeAtomicAdd(&row[p[u] + kv.w0], w[u] / nr[u]);
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) kde_scatter(kv, row, sel[u], w[u], jj[u], lane);
                }
            }
        }
    }
    lbest = wave_max(lbest);
    if (live) {
        if (lane == 0) { if (lmap) lmap[i] = (lbest > -INFINITY && ok) ? lbest : mx; if (levid) levid[i] = le; }
        kde_finalize(kv, row, ok, normalize, pdfs + i * kv.G, lane);
    }
}

}  // namespace fz
