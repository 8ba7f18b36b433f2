// agx_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the graph-build engine.
//
// All kernels are integer / byte work bound by memory latency and HBM/L2 traffic; none of it is matrix-shaped,
// so there is no MFMA here by design.  The layout rules that matter are the ones for wave64 streaming kernels:
//   * one wavefront (64 lanes) owns one tile of 64 consecutive reference positions; the only data lanes exchange
//     while a tile is swept is one DPP lane shift per hit, so there is no __syncthreads() anywhere on the hot path;
//   * per-position node buckets live in LDS as [field][variant][lane] so that every bucket access of a
//     wavefront is one conflict-free ds_read/ds_write_b32 (lane -> consecutive bank);
//   * everything a wavefront reads per hit (the tile's record stream, CIGAR runs) is wave-uniform and goes through
//     the constant address space: scalar (s_load) traffic that never waits on the per-lane loads in flight;
//   * per-lane global reads are position-consecutive across lanes (read bases, mate conti-mer lookups, node
//     keys of the successor position) and therefore coalesce.
// The per-lane algorithm itself is in agx_core.h (shared with the CPU test executor).
#include <hip/hip_runtime.h>
#include "agx_kargs.h"

#define AGX_WAVES_PER_BLOCK 4
static_assert(AGX_MAXV_LDS <= AGX_EM_W && AGX_MAXV_MID <= AGX_EM_W, "the LDS sweep writes the x -> x+1 edges of every position it finishes: its buckets must fit the edge matrix");
#define AGX_XCDS 8u                 // MI355X: 8 accelerator complex dies, 32 CUs and one L2 each
#ifndef AGX_SWEEP_WAVES
#define AGX_SWEEP_WAVES 1           // wavefronts (= consecutive tiles) per block of the node sweep.  r02 measured 0.655 / 0.656 / 0.640 / 0.779 / 0.706 ms for 1 / 2 / 4 / 6 / 8 and kept 4; r06's loop: 2.36 / 2.42 / 2.44 / - / 2.80 ms
                                    // on the 30 Mb unit — a block's LDS is free again when its LAST tile is done, and the tiles' lists differ in length
#endif

// A build queues all its kernels before the host has seen a single counter.  If the node sweeps had to give up (node pool or tile lists too
// small: the host grows them and repeats the build; a bucket beyond 64 variants: an error) parts of the node table were never written, so
// every kernel behind the sweeps first looks at the status word the sweeps leave on the device and does nothing if it is set.
#define AGX_RETURN_IF_ABORTED(word) do { if (__builtin_amdgcn_readfirstlane((int)*(word)) != 0) return; } while (0)

// ---- upload time: the conti-mer tables from their runs in two streaming kernels (agx_core.h: agx_cmseg, agx_cntrun, agx_chunk).  (r02 counted per element with atomicMax,
// scanned, filled and derived the heads — four kernels and a search per element: 1.06 ms for a 30 Mb unit.)
// one block per chunk of a count run: cm_start of its positions (+ the heads of empty positions); thread 0 of block 0 closes the table
__global__ void __launch_bounds__(256) agx_k_cm_layout(const agx_cntrun *runs, const agx_chunk *chunks, agx_u32 *cm_start, agx_cmhead *head, agx_u32 n_pos, agx_u32 n_cm) {
    const agx_chunk c = chunks[blockIdx.x];
    const agx_cntrun r = runs[c.run];
    const agx_u32 n = r.len - c.off < AGX_CM_CHUNK ? r.len - c.off : AGX_CM_CHUNK;
    for (agx_u32 j = threadIdx.x; j < n; j += 256u) agx_cm_layout_pos(r, c.off + j, cm_start, head);
    if (blockIdx.x == 0 && threadIdx.x == 0) { cm_start[n_pos] = n_cm; head[n_pos] = agx_cmhead{AGX_NONE, AGX_NONE, 0u, 0u}; }      // entry n_pos: the head of "no position"
}
// one block per chunk of a conti-mer run: keys, and the heads of the positions whose first conti-mer is here
__global__ void __launch_bounds__(256) agx_k_cm_fill(const agx_cmseg *segs, const agx_chunk *chunks, const agx_u32 *cm_start, agx_cmkey *cm, agx_cmhead *head) {
    const agx_chunk c = chunks[blockIdx.x];
    const agx_cmseg g = segs[c.run];
    const agx_u32 n = g.len - c.off < AGX_CM_CHUNK ? g.len - c.off : AGX_CM_CHUNK;
    for (agx_u32 j = threadIdx.x; j < n; j += 256u) agx_cm_fill_elem(g, c.off + j, cm_start, cm, head);
}
__global__ void __launch_bounds__(256) agx_k_zero(agx_zero_args Z) {
    agx_u32 i = blockIdx.x * 256u + threadIdx.x;
    for (int s = 0; s < 8; s++) { if (i < Z.n[s]) { Z.p[s][i] = 0u; return; } i -= Z.n[s]; }
}
// The totals the host reads next to the counter words — out[0..2] = *a, *b, *c, *sum = nodes handed out (the sum of the region counters; a
// unit has fewer than 2^32 nodes: the slices' layout is refused otherwise).  Done by block 0 of the build's last kernel.
struct agx_collect_args { agx_u32 *out; const agx_u32 *a, *b, *c, *pool_cnt; agx_u32 regions; agx_u32 *sum;
                          agx_cut_args cuts; const agx_u32 *sp_rank, *tile_side_start; const unsigned long long *sp_bits; agx_u32 n_pos; };      // (the cuts of a streamed download: agx_kargs.h)
__device__ __forceinline__ void agx_collect_block(const agx_collect_args &G) {
    __shared__ agx_u32 part[256];
    if (G.cuts.n && threadIdx.x <= G.cuts.n) {
        const agx_u32 t = threadIdx.x; const bool last = t == G.cuts.n;
        const agx_u32 w = last ? G.n_pos >> 6 : G.cuts.word[t];
        agx_u32 r = G.sp_rank[w];
        if (last && (G.n_pos & 63u)) r += (agx_u32)__popcll(G.sp_bits[w] & ((1ull << (G.n_pos & 63u)) - 1ull));
        G.cuts.cut_out[t] = r;
        G.cuts.cut_out[AGX_DL_PIECES + 1u + t] = G.tile_side_start[last ? (G.n_pos + AGX_TILE - 1u) / AGX_TILE : w];
    }
    agx_u32 t = 0;
    for (agx_u32 r = threadIdx.x; r < G.regions; r += 256u) t += G.pool_cnt[(size_t)r * AGX_REGION_PAD];
    part[threadIdx.x] = t; __syncthreads();
    for (agx_u32 o = 128; o; o >>= 1) { if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { G.out[0] = *G.a; G.out[1] = *G.b; G.out[2] = *G.c; *G.sum = part[0]; }
}
// packed base classes (2 bits per base, what crosses PCIe: agx_pack_classes2) -> vote codes (one byte per base, what the sweeps gather): 16 bases per thread;
// then the bases that are not A, C, G or T, from their list
__global__ void __launch_bounds__(256) agx_k_expand_codes(const agx_u32 *packed, uint4 *vcodes, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n16) return;
    const agx_u32 v = packed[i];
    agx_u32 w[4];
    for (int j = 0; j < 4; j++) {
        agx_u32 o = 0;
        for (int b = 0; b < 4; b++) o |= (agx_u32)agx_class_vote_code((v >> (8 * j + 2 * b)) & 3u) << (8 * b);
        w[j] = o;
    }
    vcodes[i] = make_uint4(w[0], w[1], w[2], w[3]);
}
__global__ void __launch_bounds__(256) agx_k_patch_codes(const unsigned long long *other, size_t n, agx_u8 *vcodes) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) vcodes[other[i]] = agx_class_vote_code(4u);
}

// The rows out of their upload form (agx_core.h "read rows relative to the reference"): a wavefront per block of 64 rows, a lane per row.  A lane finds its row's anchor
// among the anchor bits behind the block's first anchor, its place in the stream from the block's offset and a scan of the count bytes in front of it, and decodes the
// row into LDS (agx_row_decode: the reference's prediction sixteen bases at a time, then one byte per difference); the 64 rows — contiguous in the vote-code array —
// then leave LDS in 16-byte pieces, lane after lane.  Dynamic LDS: 64 * stride bytes (stride <= AGX_ROW_MAXSTRIDE).
__global__ void __launch_bounds__(64) agx_k_expand_rows(const agx_whit *hits, agx_u32 nh, const agx_wside *sides, const agx_wrun *runs, const agx_u32 *anchor_bits, const agx_u32 *block_first,
                                                         const agx_u8 *cnt, const agx_u32 *block_off, const agx_u16 *units, const agx_u32 *wref, agx_u8 *vcodes, agx_u32 n_rows, agx_u32 stride) {
    extern __shared__ __attribute__((aligned(16))) agx_u32 row_lds[];
    const agx_u32 b = blockIdx.x, l = threadIdx.x, row = b * 64u + l;
    const bool valid = row < n_rows;
    const agx_u32 c = cnt[row];                               // (padded with zeros to whole blocks)
    const agx_u32 u = valid ? agx_row_units(c, stride) : 0u;
    agx_u32 incl = u;
    for (agx_u32 o = 1; o < 64u; o <<= 1) { const agx_u32 t = __shfl_up(incl, o, 64); if (l >= o) incl += t; }
    if (valid) {
        const agx_u32 h = agx_anchor_select(anchor_bits, block_first[b], l);
        if (h < nh) {
            const agx_whit w = hits[h];
            const agx_wrun *left = nullptr; agx_u32 nruns = 0;
            if (c != AGX_ROW_EXPLICIT && !agx_whit_left_simple(w)) { const agx_wside sd = sides[agx_whit_side(w)]; left = runs + agx_wside_left_first(w, sd); nruns = agx_wside_left_count(w, sd); }
            agx_row_decode(wref, units + (size_t)block_off[b] + (incl - u), c, w, left, nruns, stride, (agx_u8 *)row_lds + (size_t)l * stride);
        }
    }
    __syncthreads();
    const agx_u32 rows_here = n_rows - b * 64u < 64u ? n_rows - b * 64u : 64u, words = rows_here * (stride / 4u);
    agx_u32 *dst = (agx_u32 *)(vcodes + (size_t)b * 64u * stride);
    for (agx_u32 i = l; i < words / 4u; i += 64u) ((uint4 *)dst)[i] = ((const uint4 *)row_lds)[i];
    for (agx_u32 i = (words & ~3u) + l; i < words; i += 64u) dst[i] = row_lds[i];
}

// ---- the packed upload -> the working arrays (agx_core.h "wire formats"): head of a unit's first build ---------------------------------
__global__ void __launch_bounds__(256) agx_k_expand_runs(const agx_wrun *wruns, agx_run *runs, agx_u32 n_runs) {
    const agx_u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_runs) { const agx_wrun r = wruns[i]; runs[i] = agx_run{r.q, r.t, r.n}; }
}
// 16 positions per thread: one packed word in, sixteen letters out
__global__ void __launch_bounds__(256) agx_k_expand_ref(const agx_u32 *packed, uint4 *ref, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n16) return;
    const agx_u32 v = packed[i];
    agx_u32 w[4];
    for (int j = 0; j < 4; j++) { agx_u32 o = 0; for (int b = 0; b < 4; b++) o |= (agx_u32)(agx_u8)agx_ref_base((v >> (8 * j + 2 * b)) & 3u) << (8 * b); w[j] = o; }
    ref[i] = make_uint4(w[0], w[1], w[2], w[3]);
}
// one block per stretch of other bytes
__global__ void __launch_bounds__(256) agx_k_patch_ref(const agx_refx *x, agx_u8 *ref) {
    const agx_refx r = x[blockIdx.x];
    for (agx_u32 j = threadIdx.x; j < r.len; j += 256u) ref[(size_t)r.pos + j] = (agx_u8)r.byte;
}

// ---- hit_prep: one thread per hit, hits in TILE order ------------------------------------------------------------------------------
// Thread i takes hit perm[i]: the staging sorted the hits by the tile of their first arrival (stage_order, agx_engine.cpp; the SAM file's own order is random in
// position).  Neighbouring lanes then want the same tiles: the histogram's device-scope atomics — the expensive part of binning on this chip (8 L2s: they are resolved
// behind them) — collapse to one per run of equal tiles, and the derived records leave in the order the tile lists will read them.  r04 walked the hits in file order and
// stored every (tile, hit) pair into a slot of the tile's own: 12.7 M scattered 4-byte stores per 30 Mb unit, 893 MB of write traffic for 250 MB of payload.
// The kernel still decides the one rule that needs the file order (a later hit of a pair landing on an earlier one is dropped, AG:1650-1655): it reads the hit's file
// neighbours through the wire records, which stay in file order.
__global__ void __launch_bounds__(256) agx_k_hit_prep(agx_prep_args A) {
    const agx_u32 i = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    const bool mine = i < A.n_hits;
    agx_dhit d; d.flags = AGX_HF_SKIP; d.x_lo = 1; d.x_hi = 0; d.a_nruns = 0;
    if (mine) {
        // the hit straight from its wire record (16 bytes, + 12 for the three in eight with a multi-run mate)
        const agx_whit *wh = A.whits; const agx_wside *sd = A.sides;
        const agx_u32 h = A.tiled ? i : A.perm[i];
        const agx_whit w = wh[h];
        const agx_hit H = agx_unpack_hit(w, sd);
        bool dup; agx_u32 row;
        if (A.tiled) { dup = (w.flags & AGX_WF_DUP) != 0; row = i; A.perm_out[i] = w.row; }      // (the record's row field carries the hit's number: the key the tile lists are ranked by)
        else { dup = H.back != 0 && agx_hit_dup_by([wh, sd](agx_u32 j) { return agx_unpack_hit(wh[j], sd); }, A.runs, h); row = H.slot1; }
        const int rc = agx_hit_prep(H, dup, (H.pad[0] & 1u) != 0, row, A.runs, A.k, d);      // staged hit: row of the a mate's bases
        if (rc) atomicOr(A.err, 1u);
        if (!(d.flags & AGX_HF_SKIP) && (d.x_hi >= A.n_pos || d.x_lo > d.x_hi)) { atomicOr(A.err, 2u); d.flags |= AGX_HF_SKIP; }
    }
    const bool kept = mine && !(d.flags & AGX_HF_SKIP);
    const agx_u32 t0 = kept ? d.x_lo / AGX_TILE : 0u, t1 = kept ? d.x_hi / AGX_TILE : 0u;
    // the order is the staging's claim: a hit that is not where its first tile's hits stand would silently miss its tiles' lists
    if (kept && !(A.tile_first[t0] <= i && i < A.tile_first[t0 + 1])) atomicOr(A.err, 4u);
    const bool lng = kept && t1 - t0 >= A.lookback;
    if (mine) A.ckey[i] = (kept && !lng) ? t1 : AGX_NONE;
    if (lng) { const agx_u32 at = atomicAdd(A.long_count, 1u); if (at < AGX_LONG_MAX) A.long_list[at] = i; }
    // Lanes that want the same tile as their s-th one and are neighbours form a run of equal tile numbers; the first pending lane of every run adds the run's
    // pending lanes to the tile's counter — all runs in the same atomic instruction, nothing is returned (the lists are filled by agx_k_tile_fill from the order itself).
    const unsigned long long below = (1ull << lane) - 1ull;
    for (agx_u32 s = 0; s < 4; s++) {                                        // the hit's s-th tile
        const agx_u32 t = t0 + s;
        const bool pend = kept && t <= t1;
        const agx_u32 tp = (agx_u32)__shfl_up((int)t, 1, 64);
        const unsigned long long starts = __ballot(lane == 0 || t != tp), pm = __ballot(pend);
        const agx_u32 first = 63u - (agx_u32)__builtin_clzll(starts & (below | (1ull << lane)));        // first lane of my run
        const unsigned long long after = starts & ~(below | (1ull << lane));                                 // run starts above me
        const agx_u32 end = after ? (agx_u32)__builtin_ctzll(after) : 64u;                                    // one past my run
        const unsigned long long run = (end == 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << first) - 1ull);
        const unsigned long long p = pm & run;                                                                 // pending lanes of my run
        if (pend && lane == (agx_u32)__builtin_ctzll(p)) (void)atomicAdd(&A.tile_cnt[t], (agx_u32)__popcll(p));
    }
    if (kept && t1 - t0 >= 4) for (agx_u32 t = t0 + 4; t <= t1; t++) (void)atomicAdd(&A.tile_cnt[t], 1u);      // spans more than four tiles: the rest, one by one
    if (mine) A.dhit[i] = d;
}

// ---- exclusive scan of a u32 array (three small kernels up to 16 M elements: blocks, block sums, add) --------------------
// each block scans AGX_SCAN_BLOCK elements (256 threads x AGX_SCAN_ITEMS)
#define AGX_SCAN_ITEMS 16
#define AGX_SCAN_BLOCK (256u * AGX_SCAN_ITEMS)
__global__ void __launch_bounds__(256) agx_k_scan_blocks(const agx_u32 *in, agx_u32 *out, agx_u32 *block_sums, agx_u32 n) {
    __shared__ agx_u32 sh[256];
    const agx_u32 base = blockIdx.x * AGX_SCAN_BLOCK + threadIdx.x * AGX_SCAN_ITEMS;
    agx_u32 v[AGX_SCAN_ITEMS], s = 0;
    for (int i = 0; i < AGX_SCAN_ITEMS; i++) { v[i] = (base + i < n) ? in[base + i] : 0u; s += v[i]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (agx_u32 off = 1; off < 256; off <<= 1) {
        agx_u32 t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    agx_u32 run = sh[threadIdx.x] - s;      // exclusive prefix of this thread's first element
    for (int i = 0; i < AGX_SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (threadIdx.x == 255 && block_sums) block_sums[blockIdx.x] = sh[255];
}
__global__ void __launch_bounds__(256) agx_k_scan_add(agx_u32 *out, const agx_u32 *block_offsets, agx_u32 n) {
    const agx_u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] += block_offsets[i / AGX_SCAN_BLOCK];
}

// The same scan in ONE launch (decoupled look-back): every block of 4096 elements publishes its sum in a 64-bit descriptor (flag in the top
// bits, value below), then finds its exclusive prefix by looking back over its predecessors' descriptors — a wavefront reads 64 of them at a
// time — until it meets one that already holds an inclusive prefix, and publishes its own.  desc[] must be zero when the kernel starts.
// (A command boundary costs the stream ~8 us: three launches per scan were most of what a small scan cost.)
// A block that looks back spins until its predecessors have published, so a predecessor must never be a block that cannot start: HIP does
// not promise dispatch in blockIdx order.  The grid is therefore capped at what the device holds at once (agx_scan_grid: from the occupancy
// query, halved because two build streams may scan at the same time) and workgroup g takes the 4096-element blocks g, g + grid, g + 2 grid, .. in that order: whatever a block waits
// for belongs to a workgroup that is running or has only earlier blocks to finish first.
#define AGX_SCAN_AGG (1ull << 62)
#define AGX_SCAN_PFX (2ull << 62)
#define AGX_SCAN_GRID 1024u      // upper bound; the launcher takes what the device at hand holds at once (agx_scan_grid)
__global__ void __launch_bounds__(256) agx_k_scan_lookback(const agx_u32 *in, agx_u32 *out, agx_u32 n, unsigned long long *desc, agx_u32 n_blocks) {
    __shared__ agx_u32 sh[256];
    __shared__ agx_u32 sh_excl;
    for (agx_u32 b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const agx_u32 base = b * AGX_SCAN_BLOCK + threadIdx.x * AGX_SCAN_ITEMS;
        agx_u32 v[AGX_SCAN_ITEMS], s = 0;
        for (int i = 0; i < AGX_SCAN_ITEMS; i++) { v[i] = (base + i < n) ? in[base + i] : 0u; s += v[i]; }
        __syncthreads();                                      // (the previous block's readers of sh / sh_excl are done)
        sh[threadIdx.x] = s;
        __syncthreads();
        for (agx_u32 off = 1; off < 256; off <<= 1) {
            agx_u32 t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        const agx_u32 total = sh[255];
        if (threadIdx.x < 64) {                               // the first wavefront looks back
            const agx_u32 lane = threadIdx.x;
            if (lane == 0) __hip_atomic_store(&desc[b], (b == 0 ? AGX_SCAN_PFX : AGX_SCAN_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            agx_u32 excl = 0;
            for (long long hi = (long long)b - 1; hi >= 0;) {                     // window of predecessors hi, hi-1, .., hi-63
                const long long j = hi - lane;
                unsigned long long d = AGX_SCAN_PFX;                              // lanes before block 0: a prefix of 0
                if (j >= 0) do { d = __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((d >> 62) == 0);
                const unsigned long long pfx = __ballot((d >> 62) == 2);
                const agx_u32 stop = pfx ? (agx_u32)__builtin_ctzll(pfx) : 63u;   // nearest predecessor that holds an inclusive prefix
                agx_u32 part = lane <= stop ? (agx_u32)d : 0u;
                for (agx_u32 o = 32; o; o >>= 1) part += __shfl_down(part, o, 64);
                excl += __shfl(part, 0, 64);
                if (pfx) break;
                hi -= 64;
            }
            if (lane == 0) {
                if (b) __hip_atomic_store(&desc[b], AGX_SCAN_PFX | (unsigned long long)(agx_u32)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh_excl = excl;
            }
        }
        __syncthreads();
        agx_u32 run = sh_excl + sh[threadIdx.x] - s;
        for (int i = 0; i < AGX_SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    }
}

// ---- tile lists: the 32-byte record stream the sweeps read, hits of a tile in SAM order --------------------------------------------------
#define AGX_SORT_LDS 512       // list entries of a tile sorted in LDS (a tile of the bench units holds ~27; 2048 — 32 KB per block — kept the kernel at 20 of a CU's 32 wavefronts)
// A record is the hit's LEAN record for this tile (agx_core.h: agx_lrec — its arrivals in the tile as one or two runs of lanes, or "look at dhit[hit]").  i: the hit's place in the tile order.
__device__ __forceinline__ void agx_put_lrec(uint4 *recs, size_t at, const agx_lrec &r) {
    recs[2 * at] = make_uint4(r.qoff1, r.boff1, r.qoff2, r.boff2); recs[2 * at + 1] = make_uint4(r.slot, r.lenjs, r.geo, r.hit);
}
__device__ __forceinline__ void agx_put_rec(uint4 *recs, agx_u32 at, const agx_dhit *dhit, agx_u32 i, const agx_run *runs, agx_u32 tile, agx_u32 k) {
    const uint2 *g = reinterpret_cast<const uint2 *>(dhit + i);      // (five 8-byte words, never the struct: it would live in scratch memory)
    const uint2 w0 = g[0], w1 = g[1], w2 = g[2], w3 = g[3], w4 = g[4];
    agx_put_lrec(recs, at, agx_lean_make_v(w0.x, w0.y, w1.x, w1.y, w2.x, w2.y & 0xFFFFu, w2.y >> 16, w3.x & 0xFFFFu, w3.x >> 16, w3.y, w4.x, w4.y, runs, tile, k, i));
}
// The hits are in the order of their first tile, so the hits that reach tile t are among those whose first tile is t - lookback + 1 .. t: a contiguous WINDOW of the
// order (tile_first), filtered by the last tile each hit reaches (ckey) — coalesced reads of 4-byte keys and hit numbers, no scatter, no atomics — plus the few hits that
// span more tiles than the window looks back over (long_list).  What is kept is compacted into the wavefront's low lanes, ranked by hit number (= place in the SAM file;
// unique, so an entry's rank is the number of smaller keys) and the records are written at tile_off[t] + rank.  The count must be the histogram's (hit_prep counted the
// same hits): err bit 3 otherwise.
//
// The kernel's time is the depth of its chain of dependent loads (offsets -> keys -> derived records -> runs -> stores: ~9 us per tile) times the rounds of
// resident wavefronts (a CU holds 32): 0.51 ms for the 476 k tiles of a 30 Mb unit with a wavefront per tile.  So a wavefront takes TWO tiles and issues each level
// of loads for both before it waits for either.  (Measured on the way, one tile per wavefront: a lane per window hit that fetches its own record 0.74 ms — the kept lanes
// lie scattered over the wavefront, and a vector memory instruction costs by the quads of lanes it touches: compacting the kept hits into the low lanes first is worth a
// third; the window's records staged through LDS 0.65; records padded to 48 bytes for 16-byte loads 0.80.)
// the general form, one tile: any window, any list length, long hits
__device__ __forceinline__ void agx_tile_fill_general(const agx_fill_args &A, agx_u32 tile, agx_u32 lo, agx_u32 n, agx_u32 c_lo, agx_u32 c_hi, agx_u32 n_long, agx_u32 *sh_i, agx_u32 *sh_k, agx_u32 lane) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const bool in_lds = n <= AGX_SORT_LDS;
    agx_u32 *gi = A.scratch + lo;                        // pile-ups beyond the LDS window: the kept entries' places in the order, in the tile's share of the scratch list
    agx_u32 kept = 0;
    auto take = [&](bool ok, agx_u32 i, agx_u32 h) {     // (called by all lanes together; h: the hit's number)
        const unsigned long long m = __ballot(ok);
        const agx_u32 at = kept + (agx_u32)__popcll(m & below);
        if (ok && at < n) { if (in_lds) { sh_i[at] = i; sh_k[at] = h; } else gi[at] = i; }
        kept += (agx_u32)__popcll(m);
    };
    for (agx_u32 base = c_lo; base < c_hi; base += 64) {
        const agx_u32 i = base + lane; const bool in = i < c_hi;
        const agx_u32 key = in ? A.ckey[i] : AGX_NONE, h = in ? A.perm[i] : 0u;      // (both coalesced: the hit number travels with the key instead of behind the verdict)
        take(in && key != AGX_NONE && key >= tile, i, h);
    }
    for (agx_u32 base = 0; base < n_long; base += 64) {
        const agx_u32 j = base + lane; const bool in = j < n_long;
        const agx_u32 i = in ? A.long_list[j] : 0u;
        const agx_u32 x_lo = in ? A.dhit[i].x_lo : 1u, x_hi = in ? A.dhit[i].x_hi : 0u;
        take(in && x_lo / AGX_TILE <= tile && tile <= x_hi / AGX_TILE, i, in ? A.perm[i] : 0u);
    }
    if (kept != n) { if (lane == 0) atomicOr(A.err, 8u); return; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // (single wavefront: its LDS / global writes above are visible to its own later reads)
    if (in_lds) {
        for (agx_u32 e = lane; e < n; e += 64) {
            const agx_u32 key = sh_k[e]; agx_u32 r = 0;
            for (agx_u32 j = 0; j < n; j++) r += sh_k[j] < key;
            agx_put_rec((uint4 *)A.recs, lo + r, A.dhit, sh_i[e], A.runs, tile, A.k);
        }
    } else {                                             // the same rank sort straight from L2
        for (agx_u32 e = lane; e < n; e += 64) {
            const agx_u32 i = gi[e], key = A.perm[i]; agx_u32 r = 0;
            for (agx_u32 j = 0; j < n; j++) r += A.perm[gi[j]] < key;
            agx_put_rec((uint4 *)A.recs, lo + r, A.dhit, i, A.runs, tile, A.k);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // (the wavefront's next tile uses the same LDS)
}
#ifndef AGX_FILL_TILES
#define AGX_FILL_TILES 2u      // tiles per wavefront
#endif
__global__ void __launch_bounds__(256) agx_k_tile_fill(agx_fill_args A) {
    // per wavefront 4 KB of LDS: the kept hits' places in the order [0, 512) and their hit numbers [512, 1024) (the general form); the usual tiles use 128 words each
    __shared__ agx_u32 sh[AGX_WAVES_PER_BLOCK][2 * AGX_SORT_LDS];
    static_assert(AGX_FILL_TILES * 128u <= 2u * AGX_SORT_LDS, "two tiles' kept hits must fit the wavefront's LDS");
    const agx_u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const agx_u32 t_first = __builtin_amdgcn_readfirstlane((blockIdx.x * AGX_WAVES_PER_BLOCK + wave) * AGX_FILL_TILES);
    if (t_first >= A.n_tiles) return;
    const agx_u32 n_long = __builtin_amdgcn_readfirstlane((int)*A.long_count);
    if (n_long > AGX_LONG_MAX) {                         // the fallback makes this unit's lists (agx_k_bin_fill, agx_k_tile_sort) — if it is queued; else nothing behind this kernel may run, and the host repeats the build with it
        if (!A.dense_queued && t_first == 0 && lane == 0) atomicOr(A.status, 16u);
        return;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    agx_u32 tile[AGX_FILL_TILES], lo[AGX_FILL_TILES], n[AGX_FILL_TILES], c_lo[AGX_FILL_TILES], c_hi[AGX_FILL_TILES]; bool todo[AGX_FILL_TILES], fast[AGX_FILL_TILES];
    // level 0 (scalar loads): list offsets and windows
#pragma unroll
    for (agx_u32 t = 0; t < AGX_FILL_TILES; t++) {
        tile[t] = t_first + t;
        const bool valid = tile[t] < A.n_tiles;
        const agx_u32 tt = valid ? tile[t] : t_first;
        lo[t] = agx_uload(A.tile_off, tt); const agx_u32 hi_off = agx_uload(A.tile_off, tt + 1); n[t] = hi_off - lo[t];
        c_lo[t] = agx_uload(A.tile_first, tt >= A.lookback - 1u ? tt - (A.lookback - 1u) : 0u); c_hi[t] = agx_uload(A.tile_first, tt + 1);
        todo[t] = valid && n[t] != 0 && hi_off <= A.cap;      // (lists that did not fit: the host grows them and re-runs)
        fast[t] = todo[t] && n_long == 0 && c_hi[t] - c_lo[t] <= 64u;
    }
    // level 1: keys and hit numbers of the windows (a lane per hit), both tiles' loads in flight together
    agx_u32 key[AGX_FILL_TILES], hn[AGX_FILL_TILES];
#pragma unroll
    for (agx_u32 t = 0; t < AGX_FILL_TILES; t++) {
        const agx_u32 i = c_lo[t] + lane; const bool in = fast[t] && i < c_hi[t];
        key[t] = in ? A.ckey[i] : AGX_NONE; hn[t] = in ? A.perm[i] : 0u;
    }
    // level 2: the kept hits into the low lanes (through LDS)
#pragma unroll
    for (agx_u32 t = 0; t < AGX_FILL_TILES; t++) {
        const bool ok = fast[t] && key[t] != AGX_NONE && key[t] >= tile[t];
        const unsigned long long m = __ballot(ok);
        if (fast[t] && (agx_u32)__popcll(m) != n[t]) { if (lane == 0) atomicOr(A.err, 8u); fast[t] = false; todo[t] = false; }
        if (ok && fast[t]) { const agx_u32 at = (agx_u32)__popcll(m & below); sh[wave][t * 128u + at] = c_lo[t] + lane; sh[wave][t * 128u + 64u + at] = hn[t]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // (single wavefront: its LDS writes are visible to its own later reads)
    // level 3: the derived records of the kept hits — as five 8-byte words each, never as a struct: a record that exists as a struct in registers ends up in scratch
    // memory (agx_tile_piece_v; measured: 0.72 ms instead of 0.36)
    bool act[AGX_FILL_TILES]; agx_u32 mine[AGX_FILL_TILES], place[AGX_FILL_TILES]; uint2 w[AGX_FILL_TILES][5];
#pragma unroll
    for (agx_u32 t = 0; t < AGX_FILL_TILES; t++) {
        act[t] = fast[t] && lane < n[t];
        const agx_u32 i = act[t] ? sh[wave][t * 128u + lane] : (fast[t] ? c_lo[t] : 0u); mine[t] = act[t] ? sh[wave][t * 128u + 64u + lane] : 0u; place[t] = i;
        const uint2 *g = reinterpret_cast<const uint2 *>(A.dhit + i);
#pragma unroll
        for (int q = 0; q < 5; q++) w[t][q] = g[q];
    }
    // level 4: ranks (while the records travel), level 5: the tile's pieces of the records, written at their ranks
    agx_u32 r[AGX_FILL_TILES];
#pragma unroll
    for (agx_u32 t = 0; t < AGX_FILL_TILES; t++) {
        r[t] = 0;
        if (fast[t]) for (agx_u32 j = 0; j < n[t]; j++) r[t] += sh[wave][t * 128u + 64u + j] < mine[t];
    }
    uint4 *recs = (uint4 *)A.recs;
#pragma unroll
    for (agx_u32 t = 0; t < AGX_FILL_TILES; t++)
        if (act[t]) agx_put_lrec(recs, (size_t)lo[t] + r[t], agx_lean_make_v(w[t][0].x, w[t][0].y, w[t][1].x, w[t][1].y, w[t][2].x, w[t][2].y & 0xFFFFu, w[t][2].y >> 16, w[t][3].x & 0xFFFFu, w[t][3].x >> 16,
                                                                            w[t][3].y, w[t][4].x, w[t][4].y, A.runs, tile[t], A.k, place[t]));
    // wider windows (pile-ups, long reads) and units with long hits: one tile after the other
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
    for (agx_u32 t = 0; t < AGX_FILL_TILES; t++)
        if (todo[t] && !fast[t]) agx_tile_fill_general(A, tile[t], lo[t], n[t], c_lo[t], c_hi[t], n_long, sh[wave], sh[wave] + AGX_SORT_LDS, lane);
}

// The fallback — a unit with more than AGX_LONG_MAX hits that span more tiles than the window looks back over (very long reads, many long deletions): every hit takes
// its places in dense lists from a second counter per tile (a scatter of 4-byte words, as r04 did for every unit), and a wavefront per tile rank-sorts its list.
__global__ void __launch_bounds__(256) agx_k_bin_fill(agx_bin_args A) {
    const agx_u32 i = blockIdx.x * 256u + threadIdx.x;
    if ((agx_u32)__builtin_amdgcn_readfirstlane((int)*A.long_count) <= AGX_LONG_MAX) return;      // (nothing to do: agx_k_tile_fill makes the lists)
    if (i >= A.n_hits) return;
    const agx_dhit d = A.dhit[i];
    if (d.flags & AGX_HF_SKIP) return;
    for (agx_u32 t = d.x_lo / AGX_TILE; t <= d.x_hi / AGX_TILE; t++) { const agx_u32 at = A.tile_off[t] + atomicAdd(&A.cursor[t], 1u); if (at < A.cap) A.unsorted[at] = i; }
}
__global__ void __launch_bounds__(256) agx_k_tile_sort(agx_fill_args A) {
    __shared__ agx_u32 sh_i[AGX_WAVES_PER_BLOCK][AGX_SORT_LDS], sh_k[AGX_WAVES_PER_BLOCK][AGX_SORT_LDS];
    const agx_u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const agx_u32 tile = __builtin_amdgcn_readfirstlane(blockIdx.x * AGX_WAVES_PER_BLOCK + wave);
    if (tile >= A.n_tiles) return;
    if ((agx_u32)__builtin_amdgcn_readfirstlane((int)*A.long_count) <= AGX_LONG_MAX) return;
    const agx_u32 lo = A.tile_off[tile], n = A.tile_off[tile + 1] - lo;
    if (A.tile_off[tile + 1] > A.cap) return;            // lists did not fit: the host grows them and re-runs
    const agx_u32 *src = A.scratch + lo;
    if (n <= AGX_SORT_LDS) {
        for (agx_u32 e = lane; e < n; e += 64) { const agx_u32 i = src[e]; sh_i[wave][e] = i; sh_k[wave][e] = A.perm[i]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        for (agx_u32 e = lane; e < n; e += 64) {
            const agx_u32 key = sh_k[wave][e]; agx_u32 r = 0;
            for (agx_u32 j = 0; j < n; j++) r += sh_k[wave][j] < key;
            agx_put_rec((uint4 *)A.recs, lo + r, A.dhit, sh_i[wave][e], A.runs, tile, A.k);
        }
    } else {
        for (agx_u32 e = lane; e < n; e += 64) {
            const agx_u32 i = src[e], key = A.perm[i]; agx_u32 r = 0;
            for (agx_u32 j = 0; j < n; j++) r += A.perm[src[j]] < key;
            agx_put_rec((uint4 *)A.recs, lo + r, A.dhit, i, A.runs, tile, A.k);
        }
    }
}

// ---- hit records of a tile ------------------------------------------------------------------------------------------------
// A tile's sweep reads its record list strictly in order and every lane needs every record: the list index is wave-uniform, so the
// read goes through the constant address space and becomes one s_load_dwordx8 into SGPRs (the list was written by an earlier
// kernel).  Scalar loads are counted by lgkmcnt, not vmcnt: waiting for a record never drains the per-lane global loads that the
// sweep keeps in flight one hit ahead, and most of the arrival decode runs on the scalar unit.
struct agx_tile_recs {
    const uint4 *recs; const agx_dhit *dhit;
    typedef agx_u32 v4 __attribute__((ext_vector_type(4)));
    typedef agx_u32 v2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(4))) v4 *cptr;
    typedef const __attribute__((address_space(4))) v2 *cptr2;
    // the lean record of entry i (pass 0)
    __device__ __forceinline__ agx_lrec lean(agx_u32 i) const {
        const cptr p = (cptr)(recs + 2 * (size_t)i);
        const v4 a = p[0], b = p[1];
        return agx_lrec{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    }
    // the derived record of the hit of entry i (every other reader: the wider passes decode the hit itself)
    __device__ __forceinline__ agx_dhit of_hit(agx_u32 hit) const {
        static_assert(sizeof(agx_dhit) == 40, "a derived record is ten words");
        const cptr p = (cptr)(dhit + hit); const cptr2 p2 = (cptr2)((const char *)(dhit + hit) + 32);
        const v4 a = p[0], b = p[1]; const v2 c = p2[0];
        agx_dhit d;
        d.a_t0 = a.x; d.b_t0 = a.y; d.a_runs = a.z; d.b_runs = a.w; d.a_slot = b.x;
        d.len = (agx_u16)(b.y & 0xFFFFu); d.jstar = (agx_u16)(b.y >> 16); d.a_nruns = (agx_u16)(b.z & 0xFFFFu); d.b_nruns = (agx_u16)(b.z >> 16);
        d.flags = b.w; d.x_lo = c.x; d.x_hi = c.y;
        return d;
    }
    __device__ __forceinline__ agx_dhit operator()(agx_u32 i) const {
        typedef const __attribute__((address_space(4))) agx_u32 *wptr;
        return of_hit(((wptr)(recs + 2 * (size_t)i))[7]);
    }
};

// ---- node sweep ------------------------------------------------------------------------------------------------------
// inclusive scan over the wavefront by DPP moves: inside every row of 16 lanes by shifts of 1, 2, 4, 8 (a lane the shift would fetch from outside its row reads 0), then the
// last lane of row 0 / row 2 added to all of row 1 / row 3 (row_bcast:15 on rows 1 and 3) and lane 31 to rows 2 and 3 (row_bcast:31).  Twelve vector instructions and no LDS
// round trip (r02-r05: six ds_bpermute_b32, each waited for — twice per tile, in the write-out where a wavefront has nothing else to do)
__device__ __forceinline__ agx_u32 agx_wave_incl_scan(agx_u32 v, agx_u32 lane) {
    (void)lane;
    v += (agx_u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += (agx_u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += (agx_u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += (agx_u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);      // row_shr:8
    v += (agx_u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1 and 3
    v += (agx_u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2 and 3
    return v;
}

// ---- pass 0's loop over a tile's list (r06) ----------------------------------------------------------------------------------------------
// agx_node_sweep_lane (agx_core.h, shared with the CPU executor) keeps every lane predicate as a 0/1 word: the compiler then moves them between
// VGPRs and SGPR masks all the time (v_cndmask 0,1 / v_cmp_ne 0) and the fast path of one list entry issues ~60 vector + ~60 scalar instructions.
// Here a lane predicate IS a 64-bit scalar mask (ballot in, inverse ballot out): the logic between predicates runs on the scalar unit, the
// counters take the mask as it is, the x -> x+1 edge of the straight-line case is two mask operations and a 64-bit shift instead of a DPP move
// and a multiply per entry, the k-mer string word of an arrival is only made where a variant is stored, and a LINEAR record (five entries in six)
// is decoded by six vector instructions.  What leaves the straight-line case goes through agx_arrival_slow() — the same function the shared lane
// function calls — under ONE wave-uniform branch, behind which the per-lane edge exchange of the shared function is done for that entry alone.
// Same results as agx_node_sweep_lane<true> by construction of the predicates; pinned by the GPU parity tests (node and edge tables field by field).
#ifndef AGX_SWEEP_LEAN
#define AGX_SWEEP_LEAN 1
#endif
#ifndef AGX_LEAN_VEDGE
#define AGX_LEAN_VEDGE 1          // the variant-0 edge of the straight-line case by a DPP move on the vector unit (0: three scalar mask operations — the scalar unit is the busier one: 2.47 against 2.43 ms)
#endif
typedef unsigned long long agx_m64;
#define AGX_BAL(c) ((agx_m64)__builtin_amdgcn_ballot_w64(c))
#define AGX_INV(m) (__builtin_amdgcn_inverse_ballot_w64(m))
struct agx_lbuf {                    // one buffered arrival per lane
    agx_m64 has, k1;                 // the arrival exists / counts for coverage (not K2ONLY)
    agx_u32 p0, sq, fl;              // mate position; lean record: index of the arrival's base in the stored read, general record: the whole k-mer string word (agx_sref::qlen) and, in fl, bit 0: steps to
                                     // position + 1; bit 1, all records: may not take the straight-line case (a CHAIN arrival, a step that skips positions: of all other lanes K1 <=> votes <=> steps to position + 1)
    agx_cmhead h; agx_u32 cbyte;     // as loaded
    agx_u32 slot, lenjs, geo;        // wave-uniform, from the record: read slot, read length | jstar << 16, strand and kind (agx_lrec::geo)
};
// The arrival that leaves the straight-line case, as a FUNCTION of its own (not inlined): agx_arrival_slow's nested lane-varying loops need some fifty scalar registers for their
// exec masks; inlined into the loop they pushed the loop's own masks and pointers into spill lanes and kernel-argument reloads on EVERY entry.  Behind a call the loop's state
// sits in callee-saved registers and the callee saves what it uses — only when it runs (one wave-entry in fifty: the first arrival at a position is stored without it).
// State in and out by value (registers, no stack traffic).
struct agx_slow_io { agx_u32 cnt, ok, v0_ok, v0_c0, v0_o0, v0_m, vm, sp; };
__device__ __noinline__ agx_slow_io agx_lean_slow(__attribute__((address_space(3))) agx_u32 *lds_col, const agx_cmkey *cm, int iv, agx_slow_io io, agx_u32 cx_s, agx_u32 cx_n, agx_u32 cx0_cid, agx_u32 cx0_coff,
                                                  agx_u32 p0, agx_u32 h_cid, agx_u32 h_coff, agx_u32 h_n, agx_u32 h_start, agx_u32 s0, agx_u32 s1, agx_u32 flags, agx_u32 vfield) {
    agx_sweep_args R{}; R.cm = cm; R.iv = iv;
    agx_bucket b; b.base = (agx_u32 *)lds_col; b.stride = 64; b.maxv = AGX_MAXV_LDS; b.packed = 1u;
    bool ok = io.ok != 0;
    agx_arrival_slow(R, b, io.cnt, ok, cx_s, cx_n, agx_cmkey{cx0_cid, cx0_coff}, p0, h_start, h_n, agx_cmkey{h_cid, h_coff}, s0, s1, flags & 1u, (flags >> 1) & 1u, vfield, (flags >> 2) & 1u,
                     io.v0_ok, io.v0_c0, io.v0_o0, io.v0_m, io.vm, io.sp);
    io.ok = ok ? 1u : 0u;
    return io;
}
template <class GET>
__device__ __forceinline__ bool agx_sweep_tile_lean(const agx_sweep_args &A, agx_u32 tile, agx_u32 X, const agx_bucket &b, agx_u32 &cnt, agx_u32 &pflag, agx_u32 &emask, GET get) {
    cnt = 0; pflag = 0;
    const bool live = X < A.n_pos;
    const agx_u32 lane = X & (AGX_TILE - 1u);
    bool ok = true;
    const agx_u32 lo = A.tile_off[tile], hi = A.tile_off[tile + 1];
    if (lo == hi) return true;
    if (hi - lo > 65535u) return false;                             // (packed buckets count to 65 535: the next pass takes the tile)
    const agx_u32 W = (agx_u32)(2 * A.iv + AGX_EP25), W2 = 2u * W;      // |a - b| <= W  <=>  (u32)(a - (b - W)) <= 2 W
    const agx_cmhead *cm_head = A.cm_head; const agx_u8 *vcodes = A.vcodes; const agx_u32 n_pos = A.n_pos, stride = A.stride;
    const GET recs = get;
    // variant 0's mate-side key as the shared function keeps it (v0_*), and in the form the tests below want it: offsets with the window subtracted, the window of the mate
    // position's clause per lane (everything passes where variant 0 has no mate position), the contig id with "none" replaced by an id no contig has (clause A/B passes when
    // the arrival's id is none OR differs from variant 0's: with that replacement the second test covers the first)
    // (ONLY in that form across the loop — the shared function's form is made from it where the slow path needs it: four registers less, which is what seven wavefronts per SIMD take)
    agx_u32 v0_c0x = AGX_NONE - 1u, v0_o0w = AGX_NONE - W, v0_mw = AGX_NONE - W, v0_mwin = AGX_NONE;
    agx_m64 m_v0ok = 0, m_pf1 = 0, m_e0 = 0;
    // the straight-line case's votes for variant 0: A, C, G, T, N in 6 bits each (the arrivals that take it count AND vote: coverage = the sum)
    agx_u32 acc = 0;
    auto flush = [&]() {
        // (the bucket's counters are 16-bit halves of three words — coverage | A, C | G, T | N: three adds)
        const agx_u32 ca = acc & 63u, cc = (acc >> 6) & 63u, cg = (acc >> 12) & 63u, ct = (acc >> 18) & 63u, cn = (acc >> 24) & 63u;
        agx_bucket_add<true>(agx_cnt_word(b, 0, AGX_F_COV), (ca + cc + cg + ct + cn) | (ca << 16));
        agx_bucket_add<true>(agx_cnt_word(b, 0, AGX_F_C), cc | (cg << 16));
        agx_bucket_add<true>(agx_cnt_word(b, 0, AGX_F_T), ct | (cn << 16));
        acc = 0;
    };

    auto fetch = [&](const agx_lrec &r, bool valid, agx_lbuf &P) {
        const agx_u32 g = r.geo, kind = g >> 30, L = r.lenjs & 0xFFFFu, js = r.lenjs >> 16;
        const agx_u32 rev = (g >> 24) & 1u, mrev = 0u - rev, slot = r.slot;
        agx_u32 p0, vaddr;
        if (kind == AGX_LK_ONE) {
            // one piece with mate positions and no jump (agx_lean_decode with everything else gone): lanes lo1 .. lo1 + span1, mate position lane + boff1; the lane of the K2ONLY
            // arrival and the base the stored index is counted from come digested (agx_lrec)
            const agx_m64 last = AGX_BAL(lane == r.boff2);
            P.has = valid ? AGX_BAL(lane - (g & 63u) <= ((g >> 6) & 63u)) : 0ull; P.k1 = ~last;
            m_pf1 |= P.has & ~last;
            p0 = lane + r.boff1;
            const agx_u32 stored = (lane ^ mrev) + r.qoff2;           // forward: q; reverse: L - 1 - q
            vaddr = min(stored, L - 1u);                              // (lanes outside the piece read some byte of the row)
            P.sq = stored; P.fl = 0;
        } else if (kind != AGX_LK_GENERAL) {
            // two pieces, or one with a jump at its end or without mate positions (agx_lean_decode)
            const agx_u32 lo1 = g & 63u, e1 = lo1 + ((g >> 6) & 63u), lo2 = (g >> 12) & 63u, e2 = lo2 + ((g >> 18) & 63u);
            const bool in1 = lane - lo1 <= e1 - lo1, in2 = kind == AGX_LK_TWO && lane - lo2 <= e2 - lo2, mid = (g & AGX_LF_MID) && lane > e1 && lane < lo2;
            const agx_u32 q = lane + (in2 ? r.qoff2 : r.qoff1);
            p0 = in2 ? ((g & AGX_LF_BN2) ? AGX_NONE : lane + r.boff2) : (mid || (g & AGX_LF_BN1)) ? AGX_NONE : lane + r.boff1;
            const bool lastb = q == js;
            const bool hasb = valid && (in1 || in2 || mid);
            const bool jump = hasb && !lastb && (((g & AGX_LF_JUMP1) && lane == e1) || ((g & AGX_LF_JUMP2) && in2 && lane == e2));
            const agx_m64 last = AGX_BAL(lastb);
            P.has = AGX_BAL(hasb); P.k1 = ~last;
            m_pf1 |= P.has & ~last & ~AGX_BAL(jump); pflag |= jump ? 2u : 0u;
            const agx_u32 stored = (q ^ mrev) + (L & mrev);
            vaddr = min(stored, L - 1u);
            P.sq = stored; P.fl = jump ? 2u : 0u;
        } else {
            // what does not fit a lean record: the hit's own derived record, decoded the general way
            const agx_dhit d = recs.of_hit(r.hit);
            const agx_arrival a = agx_decode_arrival(d, A.runs, X, A.k);
            const bool has = valid && live && a.has != 0;
            const bool step1 = has && a.has_succ && a.xs == X + 1, jump = has && a.has_succ && a.xs != X + 1;
            P.has = AGX_BAL(has); P.k1 = AGX_BAL(a.type != AGX_AT_K2ONLY);
            m_pf1 |= AGX_BAL(step1); pflag |= jump ? 2u : 0u;
            p0 = a.p0;
            const agx_u32 stored = rev ? L - 1u - a.q : a.q;
            vaddr = (has && a.type == AGX_AT_K1) ? stored : 0u;
            P.sq = (a.slen ? stored : 0u) | (a.slen << 16) | (rev ? 0x80000000u : 0u); P.fl = (step1 ? 1u : 0u) | ((jump || a.type == AGX_AT_CHAIN) ? 2u : 0u);
        }
        P.p0 = p0; P.slot = slot; P.lenjs = r.lenjs; P.geo = g;
        // the two loads stand once, behind the branch (addresses are chosen, not data: HISTORY.md r04).  No mate, or no arrival: the empty head at n_pos.
        P.h = cm_head[min(p0, n_pos)];
        P.cbyte = vcodes[(size_t)slot * stride + vaddr];
    };
    auto apply = [&](const agx_lbuf &P) {
        const agx_m64 cab = AGX_BAL(P.h.cid != v0_c0x) | AGX_BAL(P.h.coff - v0_o0w <= W2);                                     // agx_clause_ab against variant 0's mate-side key
        const agx_m64 cc = AGX_BAL(P.p0 == AGX_NONE) | AGX_BAL(P.p0 - v0_mw <= v0_mwin);                                       // agx_clause_c
        const agx_m64 fast = P.has & m_v0ok & cab & cc & AGX_BAL((P.h.n | (P.fl & 2u)) <= 1u);      // (at most one conti-mer at the mate position, and no lane that must not take this case)
        const agx_m64 mk = fast & P.k1;                              // of the lanes in `fast`: counts = votes = steps to position + 1
        const agx_u32 vfield = (P.cbyte >> ((P.geo >> 22) & 4u)) & 15u;      // (the strand's nibble of the vote code)
        const agx_u32 one = AGX_INV(mk) ? 1u : 0u;
        acc += one << ((vfield - (agx_u32)AGX_F_A) * 6u);           // (no LDS traffic in the straight-line case: five 6-bit counters in a register, added to the bucket every 62 entries)
#if AGX_LEAN_VEDGE
        emask |= one & (agx_u32)__builtin_amdgcn_update_dpp(0, (int)(AGX_INV(fast) ? 1u : 0u), 0x130, 0xF, 0xF, true);
#else
        m_e0 |= mk & (fast >> 1);                                    // variant 0 here -> variant 0 of the next position (lane 63's right neighbour is another tile: the edge passes')
#endif
        const agx_m64 slowm = P.has & ~fast;
        if (slowm != 0) {                                            // wave-uniform
            agx_u32 vm = AGX_INV(fast) ? 1u : 0u, sp = one;
            agx_u32 v0_ok = AGX_INV(m_v0ok) ? 1u : 0u, v0_c0 = v0_c0x == AGX_NONE - 1u ? AGX_NONE : v0_c0x, v0_o0 = v0_o0w + W, v0_m = v0_mw + W;
            if (AGX_INV(slowm)) {
                // the position's own conti-mers, fetched HERE (the index made opaque so that the load stays in this arm): four registers that only this arm reads would otherwise live
                // across the whole loop — and, at seven wavefronts per SIMD, in scratch memory: a 1 KB store per tile
                agx_u32 Xh = live ? X : n_pos; asm volatile("" : "+v"(Xh));
                const agx_cmhead hx = cm_head[Xh];
                const agx_u32 cx_s = hx.start, cx_n = hx.n; const agx_cmkey cx0 = agx_cmkey{hx.cid, hx.coff};
                const agx_u32 is_k1 = AGX_INV(P.k1) ? 1u : 0u, L = P.lenjs & 0xFFFFu, rev = (P.geo >> 24) & 1u;
                agx_u32 s1 = P.sq, vt = is_k1, st1 = is_k1 & ~(P.fl >> 1);      // (a lean record's lane: a K1 arrival votes, and steps to position + 1 unless it jumps)
                if ((P.geo >> 30) == AGX_LK_GENERAL) { vt = (is_k1 && ((P.sq >> 16) & 0x7FFFu) != 0) ? 1u : 0u; st1 = P.fl & 1u; }      // general record: a CHAIN arrival counts and does not vote (its k-mer is empty)
                else {                                               // a lean record's k-mer string word (agx_arrival_fetch): k bases from the arrival's, fewer at the read's end (the K2ONLY arrival)
                    const agx_u32 q = rev ? L - 1u - P.sq : P.sq, rest = L - q;
                    const agx_u32 slen = is_k1 ? A.k : (rest < A.k ? rest : A.k);
                    s1 = P.sq | (slen << 16) | (rev ? 0x80000000u : 0u);
                }
                if (AGX_SWEEP_FIRST && cnt == 0 && cx_n <= 1 && P.h.n <= 1) {
                    // the first arrival at a position with one candidate key (five in six of what gets here) stores variant 0 on the spot: agx_arrival_slow's first arm, written out
                    // so that only what is left pays for a call
                    agx_b(b, 0, AGX_F_CID) = cx0.cid; agx_b(b, 0, AGX_F_COFF) = cx0.coff; agx_b(b, 0, AGX_F_CID0) = P.h.cid; agx_b(b, 0, AGX_F_COFF0) = P.h.coff;
                    agx_b(b, 0, AGX_F_OFF0) = P.p0; agx_cnt_init(b, 0, is_k1);
                    agx_b(b, 0, AGX_F_S0) = P.slot; agx_b(b, 0, AGX_F_S1) = s1;
                    if (vt) agx_cnt_add<false>(b, 0, vfield, 1u);
                    cnt = 1; vm = 1u; sp = st1;
                    v0_ok = 1u; v0_c0 = P.h.cid; v0_o0 = P.h.coff; v0_m = P.p0;
                } else {
                    agx_slow_io io{cnt, ok ? 1u : 0u, v0_ok, v0_c0, v0_o0, v0_m, vm, sp};
                    io = agx_lean_slow((__attribute__((address_space(3))) agx_u32 *)b.base, A.cm, A.iv, io, cx_s, cx_n, cx0.cid, cx0.coff, P.p0, P.h.cid, P.h.coff, P.h.n, P.h.start, P.slot, s1,
                                       is_k1 | (vt << 1) | (st1 << 2), vfield);
                    cnt = io.cnt; ok = io.ok != 0; v0_ok = io.v0_ok; v0_c0 = io.v0_c0; v0_o0 = io.v0_o0; v0_m = io.v0_m; vm = io.vm; sp = io.sp;
                }
            }
            v0_c0x = v0_c0 == AGX_NONE ? AGX_NONE - 1u : v0_c0; v0_o0w = v0_o0 - W; v0_mw = v0_m - W; v0_mwin = v0_m == AGX_NONE ? AGX_NONE : W2; m_v0ok = AGX_BAL(v0_ok != 0);
            agx_edge_merge(emask, sp, (agx_u32)__builtin_amdgcn_update_dpp(0, (int)vm, 0x130, 0xF, 0xF, true));
        }
    };
    // entries lo .. hi - 1, two buffers, each refilled for the entry two places ahead as soon as it has been applied (agx_node_sweep_lane).  An index at or beyond hi reads
    // the list's LAST record again and is masked out (what lies behind a list need not be a record at all: the next tile's list may not have fitted the lists' capacity)
    const agx_u32 last = hi - 1u;
    agx_lbuf pa, pb;
    { const agx_lrec d0 = recs.lean(lo); fetch(d0, true, pa); }
    { const agx_lrec d1 = recs.lean(min(lo + 1u, last)); fetch(d1, lo + 1 < hi, pb); }
    for (agx_u32 c = lo; c < hi; ) {                                // (62 entries at a time: the register counters hold 63)
        const agx_u32 ce = hi - c > 62u ? c + 62u : hi;
        for (agx_u32 i = c; i < ce; i += 2) {
            const agx_lrec da = recs.lean(min(i + 2u, last));
            apply(pa); fetch(da, i + 2 < hi, pa);
            const agx_lrec db = recs.lean(min(i + 3u, last));
            apply(pb); fetch(db, i + 3 < hi, pb);
        }
        flush();
        c = ce + ((ce - c) & 1u);                                    // (an odd chunk — the last — has applied one masked entry beyond its end)
    }
    pflag |= AGX_INV(m_pf1) ? 1u : 0u;
    emask |= AGX_INV(m_e0) ? 1u : 0u;
    return ok;
}

template <int PASS>      // 0: every tile, AGX_MAXV_LDS variants in LDS; 1: the tiles pass 0 gave up on, AGX_MAXV_MID in LDS; 2: the rest, AGX_MAXV_BIG in global scratch;
                         // 3: what even that cannot hold, AGX_MAXV_HUGE in global scratch (only queued for units that need it)
// wavefronts per SIMD the compiler fits the kernel's registers to: pass 0 holds 5 KB of LDS per wavefront (packed buckets) — eight fit a CU's 160 KB four times over — and wants 77 VGPRs:
// six.  At seven (72 VGPRs, a 16-byte spill in front of the loop and behind the slow path's call) 2.10 against 2.14 ms; at eight (64) spills in the loop, 2.35
#ifndef AGX_SWEEP_EU
#define AGX_SWEEP_EU 7
#endif
__global__ void __launch_bounds__(64 * AGX_SWEEP_WAVES) __attribute__((amdgpu_waves_per_eu(PASS == 0 ? AGX_SWEEP_EU : 1))) agx_k_node_sweep(agx_node_kargs K) {
    constexpr bool BIG = PASS >= 2;
    constexpr agx_u32 MAXV_G = PASS == 3 ? AGX_MAXV_HUGE : AGX_MAXV_BIG;
    constexpr agx_u32 MAXV = PASS == 0 ? AGX_MAXV_LDS : AGX_MAXV_MID;
    constexpr bool PACKED = PASS == 0 && AGX_SWEEP_LEAN;          // (agx_bucket: pass 0's counters as 16-bit halves)
    __shared__ agx_u32 lds[BIG ? 1 : AGX_SWEEP_WAVES][BIG ? 1 : (PACKED ? AGX_NFP : (agx_u32)AGX_NF) * MAXV * 64];
    const agx_u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (PASS == 0 && (agx_uload(K.status, 0) & 16u)) return;      // the tile lists were not made (a unit whose long hits need the scatter fallback, which was not queued): the host repeats the build with it
    // Workgroups are handed to the 8 XCDs round-robin and every XCD has its own L2.  Neighbouring tiles read the same hit records, read
    // bases and conti-mer heads, so block b takes the (b / 8)-th block of tiles of XCD (b % 8)'s contiguous share of the unit rather than
    // tile block b: what one tile pulled into an L2 is there for its neighbours.
    agx_u32 blk = blockIdx.x;
    if (PASS == 0) { const agx_u32 share = (gridDim.x + AGX_XCDS - 1) / AGX_XCDS; blk = (blockIdx.x % AGX_XCDS) * share + blockIdx.x / AGX_XCDS; }
    const agx_u32 slot = __builtin_amdgcn_readfirstlane(blk * AGX_SWEEP_WAVES + wave);
    agx_bucket b; b.stride = 64; b.packed = PACKED ? 1u : 0u;
    if (BIG) { b.base = (PASS == 3 ? K.scratch_huge : K.scratch) + (size_t)slot * (AGX_NF * MAXV_G * 64) + lane; b.maxv = MAXV_G; }
    else { b.base = &lds[wave][lane]; b.maxv = MAXV; }
    // pass 0: one tile per wavefront.  Passes 1 and 2: a fixed set of wavefronts strides over the list of overflowed tiles.
    const agx_u32 n_work = PASS == 0 ? K.tile_hi : __builtin_amdgcn_readfirstlane(*(PASS == 1 ? K.mid_n : PASS == 2 ? K.big_n : K.huge_n));
    for (agx_u32 w = (PASS == 0 ? K.tile_lo : 0u) + slot; w < n_work; w += PASS == 0 ? 0xFFFFFFFFu : PASS == 1 ? AGX_MID_WAVES : PASS == 2 ? AGX_BIG_WAVES : AGX_HUGE_WAVES) {
        const agx_u32 tile = PASS == 0 ? w : __builtin_amdgcn_readfirstlane((PASS == 1 ? K.mid_list : PASS == 2 ? K.big_list : K.huge_list)[w]);
        if (K.S.tile_off[tile + 1] > K.list_cap) { if (lane == 0) atomicOr(K.status, 4u); return; }      // lists did not fit: nothing after the sweeps may run
        const agx_u32 X = tile * AGX_TILE + lane;
        agx_u32 cnt = 0, pflag = 0, emask = 0;
        const agx_tile_recs hits{K.S.tile_recs, K.S.dhit};
        // every lane hands the variants a hit touched to its left neighbour (agx_edge_merge): the x -> x+1 edges of 63 of the tile's 64
        // positions fall out of the sweep itself; the fallback pass leaves them to the edge passes (its buckets exceed the edge matrix)
        bool ok;
        if (PASS == 0 && AGX_SWEEP_LEAN) ok = agx_sweep_tile_lean(K.S, tile, X, b, cnt, pflag, emask, hits);
        else ok = agx_node_sweep_lane<!BIG>(K.S, tile, X, b, cnt, pflag, hits, [&](agx_u32 vm, agx_u32 sp) {
            // lane i reads lane i+1 with one DPP move (wave_shl:1; the last lane reads 0: its edges belong to the edge passes)
            if (!BIG) agx_edge_merge(emask, sp, (agx_u32)__builtin_amdgcn_update_dpp(0, (int)vm, 0x130, 0xF, 0xF, true));
        });
        if (__ballot(!ok) != 0ull) {                       // wave-uniform
            if (lane == 0) {
                if (PASS == 3 || (PASS == 2 && !K.huge_queued)) atomicOr(K.status, 2u);      // beyond every bucket that is queued: the host queues pass 3 and repeats, or gives up
                else if (PASS == 2) K.huge_list[atomicAdd(K.huge_count, 1u)] = tile;
                else if (PASS == 1) K.big_list[atomicAdd(K.big_count, 1u)] = tile;
                else { K.mid_list[atomicAdd(K.mid_count, 1u)] = tile; if (!K.fallback_queued) atomicOr(K.status, 8u); }      // nobody will sweep it again in this build
            }
            if (PASS != 0) continue; else return;
        }
        const agx_u32 incl = agx_wave_incl_scan(cnt, lane);
        const agx_u32 total = (agx_u32)__builtin_amdgcn_readlane((int)incl, 63);
        agx_u32 base = 0;
        // Node ids come from the slice of the pool that belongs to this tile's region, through the region's own counter: a single
        // counter for the whole unit is one address that every tile's returning device-scope atomic has to queue for (measured: 0.2 ms
        // of a 0.93 ms sweep).  The ids of a unit are therefore not dense; everything downstream enters the table through node_start.
        const agx_u32 region = tile / AGX_REGION_TILES;
        const agx_u32 r_lo = agx_uload(K.region_off, region), r_hi = agx_uload(K.region_off, region + 1);
        // A region whose slice is full takes its ids from the spill area behind the slices (one counter for the unit, touched by the few
        // tiles that get there): the first build of a unit has no measurement to cut the slices by, and must not need a second one.
        if (lane == 0) {
            const agx_u32 at = r_lo + atomicAdd(K.pool_cnt + (size_t)region * AGX_REGION_PAD, total);
            base = at;
            if ((unsigned long long)at + total > r_hi) {
                const unsigned long long sp = (unsigned long long)K.spill_lo + atomicAdd(K.spill_cnt, total);
                base = sp + total <= K.S.pool_cap ? (agx_u32)sp : AGX_NONE;
            }
        }
        base = (agx_u32)__builtin_amdgcn_readfirstlane((int)base);
        if (base == AGX_NONE) { if (lane == 0) atomicOr(K.status, 1u); if (PASS != 0) continue; else return; }
        const agx_u32 my_base = base + incl - cnt;
        // (the right neighbour's values by a DPP move, wave_shl:1; lane 63 reads 0 and does not use it)
        const agx_u32 nbase = (agx_u32)__builtin_amdgcn_update_dpp(0, (int)my_base, 0x130, 0xF, 0xF, true), ncnt = (agx_u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x130, 0xF, 0xF, true);
        agx_bucket bn = b; bn.base = b.base + 1;           // the next position's bucket is the next lane's column
        const bool edges = !BIG && lane < 63u && X + 1 < K.S.n_pos && cnt <= AGX_EM_W && ncnt <= AGX_EM_W;
        const agx_u32 side = agx_node_write_lane(K.S, X, b, cnt, my_base, pflag, edges, emask, bn, nbase, ncnt);
        const agx_u32 side_incl = agx_wave_incl_scan(side, lane);
        if (X < K.S.n_pos) K.S.side_pk[X] = agx_side_pack(side_incl - side, side);
        if (lane == 63u) K.S.tile_side[tile] = side_incl;
        // a multi-variant position whose x -> x+1 edges are done but which also steps elsewhere goes through pass B for those steps
        if (edges && cnt >= 2 && (pflag & 2u)) K.slow_list[atomicAdd(K.slow_count, 1u)] = X;
        if (PASS == 0) return;
    }
}

// ---- edge build ---------------------------------------------------------------------------------------------------------
// pass A (agx_edge_fast_lane) over what the node sweep could not finish: the last position of every tile (thread per tile) and all
// positions of the tiles the fallback pass wrote (wavefront per tile); slow positions are appended to pass B's list with a wave
// ballot + one atomicAdd per wavefront
__device__ __forceinline__ void agx_append_slow(const agx_edge_kargs &K, bool slow, agx_u32 X, agx_u32 lane) {
    const unsigned long long m = __ballot(slow);
    if (!m) return;
    agx_u32 base = 0;
    if (lane == 0) base = atomicAdd(K.slow_count, (agx_u32)__popcll(m));
    base = __shfl(base, 0, 64);
    if (slow) K.slow_list[base + (agx_u32)__popcll(m & ((1ull << lane) - 1ull))] = X;
}
__global__ void __launch_bounds__(256) agx_k_edge_sweep(agx_edge_kargs K, agx_u32 boundary_blocks) {
    AGX_RETURN_IF_ABORTED(K.abort);
    const agx_u32 lane = threadIdx.x & 63u;
    if (blockIdx.x < boundary_blocks) {                  // the last position of every tile: one thread each
        const agx_u32 t = blockIdx.x * 256u + threadIdx.x;
        const agx_u32 X = t * AGX_TILE + (AGX_TILE - 1u);
        bool slow = false;
        if (t < K.S.n_tiles && X < K.S.n_pos) {
            agx_u32 nb_start = 0, nb_cnt = 0;
            if (X + 1 < K.S.n_pos) { nb_start = K.S.node_start[X + 1]; nb_cnt = K.S.node_cnt[X + 1]; }
            slow = agx_edge_fast_lane(K.S, X, K.S.node_start[X], K.S.node_cnt[X], nb_start, nb_cnt);
        }
        agx_append_slow(K, slow, X, lane);
        return;
    }
    // the other blocks: all positions of the tiles that the global-scratch pass wrote, one wavefront per tile
    const agx_u32 n = __builtin_amdgcn_readfirstlane((int)*K.big_n), blk = blockIdx.x - boundary_blocks, n_blk = gridDim.x - boundary_blocks;
    for (agx_u32 w = blk * AGX_WAVES_PER_BLOCK + (threadIdx.x >> 6); w < n; w += n_blk * AGX_WAVES_PER_BLOCK) {
        const agx_u32 X = K.big_list[w] * AGX_TILE + lane;
        agx_u32 own_start = 0, own_cnt = 0;
        if (X < K.S.n_pos) { own_start = K.S.node_start[X]; own_cnt = K.S.node_cnt[X]; }
        const agx_u32 nb_start = __shfl_down(own_start, 1, 64), nb_cnt = __shfl_down(own_cnt, 1, 64);
        // (the tile's last position belongs to the boundary blocks)
        const bool slow = lane < AGX_TILE - 1u && agx_edge_fast_lane(K.S, X, own_start, own_cnt, nb_start, nb_cnt);
        agx_append_slow(K, slow, X, lane);
    }
}

// pass B: a fixed set of wavefronts strides over the slow positions; lanes = hits of the position's tile.  The position's own data
// (both buckets' stored keys, conti-mers, the allowed-edge matrix) is wave-uniform and loaded once (agx_edge_slow_ctx); a hit on the
// register path only reports which (source, target) pair it produces, the pairs are OR-ed across the wavefront with ballots and each
// distinct pair is inserted once, by its own lane.
__device__ __forceinline__ void agx_slot_insert(const agx_edge_kargs &K, agx_u32 src, agx_u32 dst) {
    agx_u32 *slots = K.S.n_next + (size_t)src * AGX_MAXE;
    // A slot only ever changes from NONE to its final value, so a plain (possibly stale) 16-byte read can prove presence;
    // only an apparent NONE needs the compare-and-swap at L2.
    const uint4 seen = *reinterpret_cast<const uint4 *>(slots);
    if (seen.x == dst || seen.y == dst || seen.z == dst || seen.w == dst) return;
    const agx_u32 sv[4] = {seen.x, seen.y, seen.z, seen.w};
    for (agx_u32 e = 0; e < AGX_MAXE; e++) {
        agx_u32 cur = sv[e];
        if (cur == AGX_NONE) { cur = atomicCAS(&slots[e], AGX_NONE, dst); if (cur == AGX_NONE) return; }
        if (cur == dst) return;
    }
    // more than AGX_MAXE distinct successors: overflow list (duplicates are removed on the host) + flag in the node's byte
    const agx_u32 i2 = atomicAdd(K.ovf_count, 1u);
    if (i2 < K.ovf_cap) K.ovf[i2] = agx_edge_ovf{src, dst};
    const size_t addr = (size_t)(K.S.n_flags + src);
    atomicOr((agx_u32 *)(addr & ~(size_t)3), (agx_u32)AGX_NF_EOVF << (8u * (agx_u32)(addr & 3)));
}

// pass J: one thread per hit whose left mate has several runs (one hit in six; the host lists them when it stages the hits: r02 looked at two words
// of every hit's derived record to find them, 0.23 ms on a 30 Mb unit); agx_edge_jump_hit drops the ones hit_prep skipped
__global__ void __launch_bounds__(256) agx_k_edge_jump(agx_edge_kargs K) {
    AGX_RETURN_IF_ABORTED(K.abort);
    const agx_u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= K.n_jump) return;
    const agx_u32 h = K.jump_list[i];                   // (a place in the tile order: where the hit's derived record is)
    if (h >= K.n_hits) return;
    const agx_dhit d = K.S.dhit[h];
    agx_edge_jump_hit(K.S, d, [&](agx_u32 src, agx_u32 dst) { agx_slot_insert(K, src, dst); });
}

__global__ void __launch_bounds__(256) agx_k_edge_slow(agx_edge_kargs K) {
    AGX_RETURN_IF_ABORTED(K.abort);
    const agx_u32 lane = threadIdx.x & 63u;
    const agx_u32 wave = __builtin_amdgcn_readfirstlane(blockIdx.x * AGX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    const agx_u32 n = __builtin_amdgcn_readfirstlane(*K.slow_count);
    const agx_u32 n_waves = gridDim.x * AGX_WAVES_PER_BLOCK;
    for (agx_u32 w = wave; w < n; w += n_waves) {
        const agx_u32 X = __builtin_amdgcn_readfirstlane(K.slow_list[w]);
        const agx_u32 tile = X / AGX_TILE;
        const agx_u32 lo = K.S.tile_off[tile], hi = K.S.tile_off[tile + 1];
        agx_slow_ctx c; agx_edge_slow_ctx(K.S, X, c);
        agx_u32 pairs = 0;
        for (agx_u32 base = lo; base < hi; base += 64) {                     // wave-uniform trip count
            const agx_u32 i = base + lane; const bool on = i < hi;
            const size_t at = on ? i : lo;
            const agx_u32 hit = K.S.tile_recs[2 * at + 1].w;                             // lanes = consecutive list entries; a lean record names its hit (agx_lrec)
            const agx_dhit d = K.S.dhit[hit];
            pairs |= agx_edge_slow_pair(K.S, c, X, d, on, [&](agx_u32 src, agx_u32 dst) { agx_slot_insert(K, src, dst); });
        }
        if (c.reg) {
            agx_u32 all = 0;                                                 // OR over the wavefront, one ballot per possible pair
            for (agx_u32 vs = 0; vs < c.n; vs++) for (agx_u32 vd = 0; vd < c.n1; vd++) {
                const agx_u32 bit = 1u << (vs * AGX_SLOW_V + vd);
                if (__ballot((pairs & bit) != 0) != 0ull) all |= bit;
            }
            if (lane < AGX_SLOW_V * AGX_SLOW_V && ((all >> lane) & 1u)) agx_slot_insert(K, c.s + lane / AGX_SLOW_V, c.s1 + lane % AGX_SLOW_V);
        }
    }
}

// ---- walk preparation: renumber surviving nodes, rewrite edges, mark forced runs (agx_core.h) -------------------------------
// One thread per position is a chain of dependent loads (node_start -> flags -> edge slots -> the targets' walk ids -> marks): with every CU full the
// kernels' time was the depth of that chain times 57 rounds of resident threads over a 30 M-position unit (r02: 0.21 + 0.64 ms).  r03: a thread takes
// AGX_WP_POS positions (a block's threads side by side in each of them: the accesses stay coalesced) and issues the loads of all of them level by level; positions
// with one variant — nearly all — are finished from those registers, the others go through the general lane function.
#ifndef AGX_WP_POS
#define AGX_WP_POS 4u
#endif
// (the first threads also mark the main ids of the chain-end positions: a_mark is complete before anything reads it)
__global__ void __launch_bounds__(256) agx_k_assign_aid(agx_compact_args A, const agx_u32 *chain_end, agx_u32 n_chain_end) {
    AGX_RETURN_IF_ABORTED(A.abort);
    const agx_u32 base = blockIdx.x * (256u * AGX_WP_POS) + threadIdx.x;
    agx_u32 s[AGX_WP_POS], n[AGX_WP_POS], fl[AGX_WP_POS];
#pragma unroll
    for (agx_u32 j = 0; j < AGX_WP_POS; j++) {
        const agx_u32 X = base + j * 256u;
        if (X < n_chain_end) A.a_mark[chain_end[X]] = 1;
        const bool in = X < A.n_pos;
        s[j] = in ? A.node_start[X] : 0u; n[j] = in ? A.node_cnt[X] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (agx_u32 j = 0; j < AGX_WP_POS; j++) fl[j] = n[j] == 1u ? A.n_flags[s[j]] : 0u;
#pragma unroll
    for (agx_u32 j = 0; j < AGX_WP_POS; j++) {
        const agx_u32 X = base + j * 256u;
        if (n[j] == 0xFFFFFFFFu) continue;
        if (n[j] == 1u && !(fl[j] & AGX_NF_DEAD)) A.aid_of[s[j]] = X;                      // the position's only variant, alive: its main id
        else if (n[j] <= 1u) {                                                           // no alive variant here
            if (n[j]) A.aid_of[s[j]] = AGX_NONE;
            A.a_meta[X] = (agx_u8)(AGX_WM_ABSENT | (n[j] ? AGX_WM_ANY : 0)); A.a_str[X] = 'N'; A.a_nid[X] = AGX_NONE;
        } else agx_assign_aid_pos(A, X);
    }
}
// (the first threads also rewrite the overflow edges)
__global__ void __launch_bounds__(256) agx_k_emit_alive(agx_compact_args A, const agx_u32 *n_ovf_dev, agx_u32 ovf_cap) {
    AGX_RETURN_IF_ABORTED(A.abort);
    const agx_u32 base = blockIdx.x * (256u * AGX_WP_POS) + threadIdx.x;
    const agx_u32 n_ovf = *n_ovf_dev; A.n_ovf = n_ovf < ovf_cap ? n_ovf : ovf_cap;
    agx_u32 s[AGX_WP_POS], n[AGX_WP_POS], a[AGX_WP_POS], fl[AGX_WP_POS], pk[AGX_WP_POS]; char rf[AGX_WP_POS], bs[AGX_WP_POS]; uint4 nx[AGX_WP_POS];
#pragma unroll
    for (agx_u32 j = 0; j < AGX_WP_POS; j++) {
        const agx_u32 X = base + j * 256u;
        agx_emit_alive_ovf(A, X);
        const bool in = X < A.n_pos;
        s[j] = in ? A.node_start[X] : 0u; n[j] = in ? A.node_cnt[X] : 0u; pk[j] = in ? A.side_pk[X] : 0u; rf[j] = in ? A.ref[X] : 'N';
    }
#pragma unroll
    for (agx_u32 j = 0; j < AGX_WP_POS; j++) {
        const bool one = n[j] == 1u;
        a[j] = one ? A.aid_of[s[j]] : AGX_NONE; fl[j] = one ? A.n_flags[s[j]] : 0u; bs[j] = one ? (char)A.n_base[s[j]] : 'X';
        nx[j] = one ? *reinterpret_cast<const uint4 *>(A.n_next + (size_t)s[j] * AGX_MAXE) : make_uint4(AGX_NONE, AGX_NONE, AGX_NONE, AGX_NONE);
    }
    agx_u32 ta[AGX_WP_POS][AGX_MAXE];
#pragma unroll
    for (agx_u32 j = 0; j < AGX_WP_POS; j++) {
        const agx_u32 t[AGX_MAXE] = {nx[j].x, nx[j].y, nx[j].z, nx[j].w};
        bool open = a[j] != AGX_NONE;                     // (the slots are filled front to back: the first NONE ends the list)
#pragma unroll
        for (agx_u32 e = 0; e < AGX_MAXE; e++) { open = open && t[e] != AGX_NONE; ta[j][e] = open ? A.aid_of[t[e]] : AGX_NONE; }
    }
#pragma unroll
    for (agx_u32 j = 0; j < AGX_WP_POS; j++) {
        const agx_u32 X = base + j * 256u;
        if (n[j] > 1u) { agx_emit_alive_pos(A, X); continue; }
        if (n[j] == 0u || a[j] == AGX_NONE) continue;
        // agx_emit_alive_node() for the one variant of X, from what was loaded above
        const agx_u32 v = s[j], id = a[j];
        A.a_str[id] = bs[j] != 'X' ? bs[j] : rf[j];
        A.a_nid[id] = v;
        agx_u32 next[AGX_MAXE]; agx_u32 k = 0;
#pragma unroll
        for (agx_u32 e = 0; e < AGX_MAXE; e++) if (ta[j][e] != AGX_NONE) next[k++] = ta[j][e];
        const bool cont = k == 1 && !(fl[j] & AGX_NF_EOVF) && next[0] == id + 1;
        agx_u8 m = (agx_u8)((cont ? AGX_WM_CONT : 0) | ((fl[j] & AGX_NF_CONTIG) ? AGX_WM_CONTIG : 0));
        if (id < A.n_pos) m |= (agx_u8)(AGX_WM_ANY | ((pk[j] >> 16) ? AGX_WM_SIDE : 0));
        else A.side_xpos[id - A.n_pos] = X;
        A.a_meta[id] = m;
        if (!cont) for (agx_u32 e = 0; e < k; e++) A.a_mark[next[e]] = 1;      // racing stores of the same value
    }
}
// the special-id bitmap and its popcounts (input of the rank scan).  A wavefront takes four 64-id words — every lane one id of each, so that their loads
// are in flight together — and writes nothing for words past the live ids: sp_bits / sp_cnt are zeroed at the start of the build (the grid covers the id
// CAPACITY, 2.4 x the live ids of a first build: r02 spent 0.42 ms here on a 30 Mb unit, most of it rounds of threads that only found out they were idle)
#define AGX_SB_WORDS 4u
__global__ void __launch_bounds__(256) agx_k_special_bits(agx_compact_args A, agx_u32 n_words) {
    AGX_RETURN_IF_ABORTED(A.abort);
    A.n_ids = A.n_pos + A.tile_side_start[(A.n_pos + AGX_TILE - 1) / AGX_TILE];
    const agx_u32 lane = threadIdx.x & 63u;
    const agx_u32 w0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4u + (threadIdx.x >> 6)) * AGX_SB_WORDS);
    if (w0 >= n_words || (unsigned long long)w0 * 64u >= A.n_ids) return;          // wave-uniform
    bool f[AGX_SB_WORDS];
#pragma unroll
    for (agx_u32 j = 0; j < AGX_SB_WORDS; j++) f[j] = agx_special_id(A, (w0 + j) * 64u + lane);
#pragma unroll
    for (agx_u32 j = 0; j < AGX_SB_WORDS; j++) {
        const unsigned long long bits = __ballot(f[j]);
        if (lane == 0 && w0 + j < n_words) { A.sp_bits[w0 + j] = bits; A.sp_cnt[w0 + j] = (agx_u32)__popcll(bits); }
    }
}
// gather the special records in id order.  A wavefront takes AGX_SE_WORDS 64-id words, a lane one id of each: one id in thirteen is special, so a lane that served a single
// word had a load in flight one time in thirteen and the kernel's time was the depth of its chain (a_nid -> node fields -> the targets' walk ids -> the run of the position)
// times the rounds of resident wavefronts.  Here the loads of the four ids go out level by level (r03 did the same for the walk preparation's other kernels).
#define AGX_SE_WORDS 4u
__global__ void __launch_bounds__(256) agx_k_special_emit(agx_compact_args A, agx_u32 n_words, agx_collect_args G) {
    if (blockIdx.x == 0) agx_collect_block(G);          // (also when the build was aborted: the host sizes the retry from these)
    AGX_RETURN_IF_ABORTED(A.abort);
    const agx_u32 lane = threadIdx.x & 63u;
    const agx_u32 w0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4u + (threadIdx.x >> 6)) * AGX_SE_WORDS);
    if (w0 >= n_words) return;
    unsigned long long bits[AGX_SE_WORDS]; unsigned long long any = 0;
#pragma unroll
    for (agx_u32 j = 0; j < AGX_SE_WORDS; j++) { bits[j] = w0 + j < n_words ? A.sp_bits[w0 + j] : 0ull; any |= bits[j]; }
    if (!any) return;                                                         // wave-uniform
    bool on[AGX_SE_WORDS]; agx_u32 a[AGX_SE_WORDS], x[AGX_SE_WORDS], v[AGX_SE_WORDS], at[AGX_SE_WORDS];
    // level 1: the id's position (side ids: from the side table) and node
#pragma unroll
    for (agx_u32 j = 0; j < AGX_SE_WORDS; j++) {
        a[j] = (w0 + j) * 64u + lane; on[j] = ((bits[j] >> lane) & 1ull) != 0;
        x[j] = on[j] ? (a[j] < A.n_pos ? a[j] : A.side_xpos[a[j] - A.n_pos]) : 0xFFFFFFFFu;
        v[j] = on[j] ? A.a_nid[a[j]] : AGX_NONE;
        at[j] = on[j] ? A.sp_rank[w0 + j] + (agx_u32)__popcll(bits[j] & ((1ull << lane) - 1ull)) : AGX_NONE;
        if (at[j] >= A.sp_cap) on[j] = false;                                 // table too small: the host sees the count and repeats the build
    }
    // level 2: the node's fields; the conti-mer count of its position; the index entry of the word's lowest position
    agx_walknode wn[AGX_SE_WORDS]; uint4 nx[AGX_SE_WORDS]; agx_u32 cm_n[AGX_SE_WORDS], s0[AGX_SE_WORDS];
#pragma unroll
    for (agx_u32 j = 0; j < AGX_SE_WORDS; j++) {
        const bool node = on[j] && v[j] != AGX_NONE;
        const agx_u32 vv = node ? v[j] : 0u;
        wn[j].off0 = node ? A.nk_off0[vv] : AGX_NONE; wn[j].xpos = x[j]; wn[j].sref = node ? A.n_sref[vv] : agx_sref{0, 0};
        static_assert(AGX_MAXE == 4, "a node's edge slots are read as one uint4 (the arena's 256-byte alignment keeps d_next 16-byte aligned)");
        nx[j] = node ? *reinterpret_cast<const uint4 *>(A.n_next + (size_t)vv * AGX_MAXE) : make_uint4(AGX_NONE, AGX_NONE, AGX_NONE, AGX_NONE);
        cm_n[j] = (on[j] && A.n_seg0) ? A.cm_start[x[j] + 1] - A.cm_start[x[j]] : 0u;
        // The hop entry of a position comes from the rank-0 run that holds it.  The ids of a word are 64 neighbours — main ids are positions, side ids are in position
        // order —, so one look into the host's index of the runs (an entry per 1024 positions) for the word's lowest position, and every lane walks forward from there.
        agx_u32 xmin = x[j];
        for (agx_u32 o = 32; o; o >>= 1) { const agx_u32 t = (agx_u32)__shfl_xor((int)xmin, (int)o, 64); xmin = t < xmin ? t : xmin; }
        s0[j] = A.n_seg0 && xmin != 0xFFFFFFFFu ? agx_uload(A.seg_index, xmin / AGX_SEG_INDEX) : 0u;
    }
    // level 3: the targets' walk ids (the slots are filled front to back: the first NONE ends the list; pruned targets are dropped)
#pragma unroll
    for (agx_u32 j = 0; j < AGX_SE_WORDS; j++) {
        const agx_u32 t[AGX_MAXE] = {nx[j].x, nx[j].y, nx[j].z, nx[j].w};
        agx_u32 ta[AGX_MAXE]; bool open = true;
#pragma unroll
        for (agx_u32 e = 0; e < AGX_MAXE; e++) { open = open && t[e] != AGX_NONE; ta[e] = open ? A.aid_of[t[e]] : AGX_NONE; }
        agx_u32 k = 0;
#pragma unroll
        for (agx_u32 e = 0; e < AGX_MAXE; e++) wn[j].next[e] = AGX_NONE;
#pragma unroll
        for (agx_u32 e = 0; e < AGX_MAXE; e++) if (ta[e] != AGX_NONE) { for (agx_u32 q = 0; q < AGX_MAXE; q++) if (q == k) wn[j].next[q] = ta[e]; k++; }
    }
    // level 4: hop entries, then the stores
#pragma unroll
    for (agx_u32 j = 0; j < AGX_SE_WORDS; j++) {
        if (!on[j]) continue;
        A.sp_node[at[j]] = wn[j];
        agx_hop h; h.str_off = 0; h.len = 0; h.end_pos = 0;
        if (cm_n[j] == 1u) {
            const agx_u32 xx = x[j];
            agx_u32 si = s0[j], steps = 0;
            while (si + 1 < A.n_seg0 && A.segs[si + 1].pos0 <= xx && steps < 16u) { si++; steps++; }      // last rank-0 run with pos0 <= x: a few steps behind the index entry
            if (steps == 16u) { agx_u32 lo = si, hi = A.n_seg0; while (hi - lo > 1) { const agx_u32 mid = lo + (hi - lo) / 2; if (A.segs[mid].pos0 <= xx) lo = mid; else hi = mid; } si = lo; }      // (the one word that holds the last main ids and the first side ids)
            const agx_cmseg g = A.segs[si];
            const agx_u32 jj = xx - g.pos0;
            if (xx >= g.pos0 && jj < g.len && jj < g.hop_len0) { h.str_off = g.hop_str0 + jj; h.len = g.hop_len0 - jj; h.end_pos = g.hop_end; }
        }
        A.sp_hop[at[j]] = h;
    }
}

// records of the walk ids first + r*stride + c (r < rows, c < width), row-major into out: what the walk fetches one by one
__global__ void __launch_bounds__(256) agx_k_fetch_records(agx_compact_args A, agx_u32 first, agx_u32 stride, agx_u32 rows, agx_u32 width, agx_walknode *out) {
    const agx_u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= rows * width) return;
    out[i] = agx_walk_record(A, first + (i / width) * stride + i % width);
}

// ---- HBM -> pinned host memory by a kernel ------------------------------------------------------------------------------------
// For the few counter words a build hands to the host: they leave with the last command of the build's own chain instead of a copy on another
// stream.  (It also carried the downloads for a while; see do_download for why it no longer does.)  Up to four (dst, src, 16-byte words) segments per launch.
typedef agx_u32 agx_v4 __attribute__((ext_vector_type(4)));
struct agx_copy_args { agx_v4 *dst[4]; const agx_v4 *src[4]; unsigned long long n16[4]; };
__global__ void __launch_bounds__(256) agx_k_copy_out(agx_copy_args C) {
    for (int s = 0; s < 4; s++)
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < C.n16[s]; i += (unsigned long long)gridDim.x * 256u)
            __builtin_nontemporal_store(__builtin_nontemporal_load(&C.src[s][i]), &C.dst[s][i]);
}

// ---- host-callable launchers (kept in this translation unit so that the engine is plain C++) -------------------------
extern "C" {

void agx_launch_cm_tables(const void *cnt_runs, const void *cnt_chunks, agx_u32 n_cnt_chunks, const agx_cmseg *segs, const void *seg_chunks, agx_u32 n_seg_chunks,
                           agx_u32 *cm_start, agx_cmkey *cm, agx_cmhead *head, agx_u32 n_pos, agx_u32 n_cm, hipStream_t st) {
    if (n_cnt_chunks) hipLaunchKernelGGL(agx_k_cm_layout, dim3(n_cnt_chunks), dim3(256), 0, st, (const agx_cntrun *)cnt_runs, (const agx_chunk *)cnt_chunks, cm_start, head, n_pos, n_cm);
    if (n_seg_chunks) hipLaunchKernelGGL(agx_k_cm_fill, dim3(n_seg_chunks), dim3(256), 0, st, segs, (const agx_chunk *)seg_chunks, cm_start, cm, head);
}
void agx_launch_zero(const agx_zero_args *Z, hipStream_t st) {
    unsigned long long total = 0; for (int s = 0; s < 8; s++) total += Z->n[s];
    if (total) hipLaunchKernelGGL(agx_k_zero, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *Z);
}
void agx_launch_expand_codes(const void *packed, void *vcodes, size_t n_bases16, const unsigned long long *other, size_t n_other, hipStream_t st) {
    const size_t n16 = n_bases16 / 16;
    if (n16) hipLaunchKernelGGL(agx_k_expand_codes, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (const agx_u32 *)packed, (uint4 *)vcodes, n16);
    if (n_other) hipLaunchKernelGGL(agx_k_patch_codes, dim3((unsigned)((n_other + 255) / 256)), dim3(256), 0, st, other, n_other, (agx_u8 *)vcodes);
}
void agx_launch_patch_codes(const unsigned long long *other, size_t n_other, void *vcodes, hipStream_t st) {
    if (n_other) hipLaunchKernelGGL(agx_k_patch_codes, dim3((unsigned)((n_other + 255) / 256)), dim3(256), 0, st, other, n_other, (agx_u8 *)vcodes);
}
void agx_launch_expand_rows(const void *whits, agx_u32 nh, const void *wsides, const void *wruns, const agx_u32 *anchor_bits, const agx_u32 *block_first, const agx_u8 *cnt, const agx_u32 *block_off,
                            const agx_u16 *units, const void *wref, void *vcodes, agx_u32 n_rows, agx_u32 stride, const unsigned long long *other, size_t n_other, hipStream_t st) {
    if (n_rows) hipLaunchKernelGGL(agx_k_expand_rows, dim3((n_rows + 63u) / 64u), dim3(64), 64u * stride, st, (const agx_whit *)whits, nh, (const agx_wside *)wsides, (const agx_wrun *)wruns, anchor_bits, block_first,
                                   cnt, block_off, units, (const agx_u32 *)wref, (agx_u8 *)vcodes, n_rows, stride);
    if (n_other) hipLaunchKernelGGL(agx_k_patch_codes, dim3((unsigned)((n_other + 255) / 256)), dim3(256), 0, st, other, n_other, (agx_u8 *)vcodes);
}
void agx_launch_expand_runs(const void *wruns, agx_run *runs, agx_u32 n_runs, hipStream_t st) {
    if (n_runs) hipLaunchKernelGGL(agx_k_expand_runs, dim3((n_runs + 255) / 256), dim3(256), 0, st, (const agx_wrun *)wruns, runs, n_runs);
}
void agx_launch_expand_ref(const void *packed, void *ref, size_t n_pos16, const void *refx, agx_u32 n_refx, hipStream_t st) {
    const size_t n16 = n_pos16 / 16;
    if (n16) hipLaunchKernelGGL(agx_k_expand_ref, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (const agx_u32 *)packed, (uint4 *)ref, n16);
    if (n_refx) hipLaunchKernelGGL(agx_k_patch_ref, dim3(n_refx), dim3(256), 0, st, (const agx_refx *)refx, (agx_u8 *)ref);
}
void agx_launch_hit_prep(const agx_prep_args *A, hipStream_t st) {
    if (A->n_hits) hipLaunchKernelGGL(agx_k_hit_prep, dim3((A->n_hits + 255) / 256), dim3(256), 0, st, *A);
}

// exclusive scan of in[0..n) into out[0..n]; out[n] = total.  tmp must hold 2*(ceil(n/4096)+1) + 2*(ceil(n/4096^2)+1) + 8 words (the engine sizes it for blocks of 1024).
void agx_launch_exclusive_scan(const agx_u32 *in, agx_u32 *out, agx_u32 n, agx_u32 *tmp, hipStream_t st) {
    // scan n+1 elements (a trailing zero-extended element gives the total in out[n]); callers allocate in with n+1 entries, in[n]=0
    const agx_u32 m = n + 1;
    const agx_u32 nb = (m + AGX_SCAN_BLOCK - 1) / AGX_SCAN_BLOCK;
    if (nb == 1) { hipLaunchKernelGGL(agx_k_scan_blocks, dim3(1), dim3(256), 0, st, in, out, (agx_u32 *)nullptr, m); return; }
    agx_u32 *sums = tmp, *sums_scanned = tmp + nb + 1;
    hipLaunchKernelGGL(agx_k_scan_blocks, dim3(nb), dim3(256), 0, st, in, out, sums, m);
    // scan the block sums (recursively; nb <= 2^22 for 2^32 elements, two levels are enough in practice)
    const agx_u32 nb2 = (nb + AGX_SCAN_BLOCK - 1) / AGX_SCAN_BLOCK;
    if (nb2 == 1) hipLaunchKernelGGL(agx_k_scan_blocks, dim3(1), dim3(256), 0, st, sums, sums_scanned, (agx_u32 *)nullptr, nb);
    else {
        agx_u32 *sums2 = sums_scanned + nb + 1, *sums2_scanned = sums2 + nb2 + 1;
        hipLaunchKernelGGL(agx_k_scan_blocks, dim3(nb2), dim3(256), 0, st, sums, sums_scanned, sums2, nb);
        hipLaunchKernelGGL(agx_k_scan_blocks, dim3(1), dim3(256), 0, st, sums2, sums2_scanned, (agx_u32 *)nullptr, nb2);   // nb2 <= 4096 for n < 2^32
        hipLaunchKernelGGL(agx_k_scan_add, dim3((nb + 255) / 256), dim3(256), 0, st, sums_scanned, sums2_scanned, nb);
    }
    hipLaunchKernelGGL(agx_k_scan_add, dim3((m + 255) / 256), dim3(256), 0, st, out, sums_scanned, m);
}

// the one-launch form; desc: ceil((n+1)/4096) zeroed 64-bit words
// Workgroups of the one-launch scan: every one of them must be resident for the look-back to be safe (see agx_k_scan_lookback), also on a partition of
// the chip (CPX: 32 CUs) and while a second scan runs on the device's other build stream: half of what the occupancy query says the device
// holds, per device, at most AGX_SCAN_GRID.
static agx_u32 agx_scan_grid() {
    static agx_u32 grid[64];      // 0 = not asked yet
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return 64u;
    agx_u32 &g = grid[dev & 63];
    if (!g) {
        int per_cu = 0; hipDeviceProp_t prop; agx_u32 v = 64u;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, agx_k_scan_lookback, 256, 0) == hipSuccess && per_cu > 0 && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            v = (agx_u32)per_cu * (agx_u32)prop.multiProcessorCount / 2u;
        g = v < 16u ? 16u : v > AGX_SCAN_GRID ? AGX_SCAN_GRID : v;
    }
    return g;
}
void agx_launch_exclusive_scan1(const agx_u32 *in, agx_u32 *out, agx_u32 n, unsigned long long *desc, hipStream_t st) {
    const agx_u32 m = n + 1;                              // callers allocate in with n+1 entries, in[n] = 0: out[n] = total
    const agx_u32 nb = (m + AGX_SCAN_BLOCK - 1) / AGX_SCAN_BLOCK, grid = agx_scan_grid();
    hipLaunchKernelGGL(agx_k_scan_lookback, dim3(nb < grid ? nb : grid), dim3(256), 0, st, in, out, m, desc, nb);
}

void agx_launch_bin_fill(const agx_bin_args *A, hipStream_t st) {
    if (A->n_hits) hipLaunchKernelGGL(agx_k_bin_fill, dim3((A->n_hits + 255) / 256), dim3(256), 0, st, *A);
}
void agx_launch_tile_fill(const agx_fill_args *A, hipStream_t st) {
    const agx_u32 per_block = AGX_WAVES_PER_BLOCK * AGX_FILL_TILES;
    if (A->n_tiles) hipLaunchKernelGGL(agx_k_tile_fill, dim3((A->n_tiles + per_block - 1) / per_block), dim3(256), 0, st, *A);
}
void agx_launch_tile_sort(const agx_fill_args *A, hipStream_t st) {
    if (A->n_tiles) hipLaunchKernelGGL(agx_k_tile_sort, dim3((A->n_tiles + AGX_WAVES_PER_BLOCK - 1) / AGX_WAVES_PER_BLOCK), dim3(256), 0, st, *A);
}
void agx_launch_node_sweep(const agx_node_kargs *K, hipStream_t st) {
    const agx_u32 n = K->tile_hi > K->tile_lo ? K->tile_hi - K->tile_lo : 0u;
    // a multiple of the XCD count so that every XCD's share has the same number of blocks (blocks past the last tile do nothing)
    const agx_u32 nb = (n + AGX_SWEEP_WAVES - 1) / AGX_SWEEP_WAVES, grid = (nb + AGX_XCDS - 1) / AGX_XCDS * AGX_XCDS;
    if (n) hipLaunchKernelGGL(agx_k_node_sweep<0>, dim3(grid), dim3(64 * AGX_SWEEP_WAVES), 0, st, *K);
}
void agx_launch_node_sweep_big(const agx_node_kargs *K, hipStream_t st) {
    hipLaunchKernelGGL(agx_k_node_sweep<1>, dim3(AGX_MID_WAVES / AGX_SWEEP_WAVES), dim3(64 * AGX_SWEEP_WAVES), 0, st, *K);
    hipLaunchKernelGGL(agx_k_node_sweep<2>, dim3(AGX_BIG_WAVES / AGX_SWEEP_WAVES), dim3(64 * AGX_SWEEP_WAVES), 0, st, *K);
}
void agx_launch_node_sweep_huge(const agx_node_kargs *K, hipStream_t st) {
    hipLaunchKernelGGL(agx_k_node_sweep<3>, dim3(AGX_HUGE_WAVES / AGX_SWEEP_WAVES), dim3(64 * AGX_SWEEP_WAVES), 0, st, *K);
}
void agx_launch_edge_sweep(const agx_edge_kargs *K, hipStream_t st) {
    const agx_u32 n = K->S.n_tiles;
    if (!n) return;
    const agx_u32 nb = (n + 255) / 256;
    hipLaunchKernelGGL(agx_k_edge_sweep, dim3(nb + AGX_BIG_WAVES / AGX_WAVES_PER_BLOCK), dim3(256), 0, st, *K, nb);
}
void agx_launch_edge_jump(const agx_edge_kargs *K, hipStream_t st) {
    if (K->S.n_pos && K->n_jump) hipLaunchKernelGGL(agx_k_edge_jump, dim3((K->n_jump + 255u) / 256u), dim3(256), 0, st, *K);
}
void agx_launch_edge_slow(const agx_edge_kargs *K, hipStream_t st) {
    // persistent wavefronts: exactly as many blocks as the device holds at once (a second, partial round of blocks would idle most CUs)
    static const int blocks = [] {          // (initialised once, thread-safe: builds are queued from several host threads)
        int per_cu = 0, dev = 0; hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, agx_k_edge_slow, 256, 0) == hipSuccess && per_cu > 0 &&
            hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) return per_cu * prop.multiProcessorCount;
        return (int)(AGX_SLOW_WAVES / AGX_WAVES_PER_BLOCK);
    }();
    if (K->S.n_tiles) hipLaunchKernelGGL(agx_k_edge_slow, dim3((unsigned)blocks), dim3(256), 0, st, *K);
}

// dst: device-visible addresses of pinned host buffers; bytes are rounded up to 16 (the buffers are padded)
void agx_launch_copy_out(void *const *dst, const void *const *src, const size_t *bytes, int n, hipStream_t st) {
    for (int at = 0; at < n; at += 4) {
        agx_copy_args C; unsigned long long total = 0;
        for (int s = 0; s < 4; s++) { const bool on = at + s < n; C.dst[s] = on ? (agx_v4 *)dst[at + s] : nullptr; C.src[s] = on ? (const agx_v4 *)src[at + s] : nullptr; C.n16[s] = on ? (bytes[at + s] + 15) / 16 : 0; total += C.n16[s]; }
        if (!total) continue;
        const unsigned long long want = (total + 256ull * 8 - 1) / (256ull * 8);       // ~8 words per thread
        hipLaunchKernelGGL(agx_k_copy_out, dim3((unsigned)(want < 1 ? 1 : want > 512 ? 512 : want)), dim3(256), 0, st, C);
    }
}
void agx_launch_fetch_records(const agx_compact_args *A, agx_u32 first, agx_u32 stride, agx_u32 rows, agx_u32 width, agx_walknode *out, hipStream_t st) {
    const agx_u32 n = rows * width;
    if (n) hipLaunchKernelGGL(agx_k_fetch_records, dim3((n + 255) / 256), dim3(256), 0, st, *A, first, stride, rows, width, out);
}
void agx_launch_compact(const agx_compact_args *A, const agx_u32 *chain_end, agx_u32 n_chain_end, const agx_u32 *n_ovf_dev, agx_u32 ovf_cap, hipStream_t st) {
    const agx_u32 n1 = A->n_pos > n_chain_end ? A->n_pos : n_chain_end, n2 = A->n_pos > ovf_cap ? A->n_pos : ovf_cap;
    if (n1) hipLaunchKernelGGL(agx_k_assign_aid, dim3((n1 + 256 * AGX_WP_POS - 1) / (256 * AGX_WP_POS)), dim3(256), 0, st, *A, chain_end, n_chain_end);
    if (n2) hipLaunchKernelGGL(agx_k_emit_alive, dim3((n2 + 256 * AGX_WP_POS - 1) / (256 * AGX_WP_POS)), dim3(256), 0, st, *A, n_ovf_dev, ovf_cap);
}
// sparse record table over n_words 64-id words (the id capacity; the live id count is read on the device); scan_tmp as for the scans
void agx_launch_special(const agx_compact_args *A, agx_u32 n_words, agx_u32 *sp_rank, agx_u32 *scan_tmp, unsigned long long *desc,
                        agx_u32 *out, const agx_u32 *a, const agx_u32 *b, const agx_u32 *pool_cnt, agx_u32 regions, agx_u32 *sum, const agx_cut_args *cuts, hipStream_t st) {
    agx_collect_args G{out, a, b, sp_rank + n_words, pool_cnt, regions, sum, agx_cut_args{}, sp_rank, A->tile_side_start, A->sp_bits, A->n_pos};
    if (cuts) G.cuts = *cuts; else G.cuts.n = 0;
    const agx_u32 blocks = (n_words + 4u * AGX_SE_WORDS - 1u) / (4u * AGX_SE_WORDS);
    hipLaunchKernelGGL(agx_k_special_bits, dim3((n_words + 4u * AGX_SB_WORDS - 1u) / (4u * AGX_SB_WORDS)), dim3(256), 0, st, *A, n_words);
    if (desc) agx_launch_exclusive_scan1(A->sp_cnt, sp_rank, n_words, desc, st); else agx_launch_exclusive_scan(A->sp_cnt, sp_rank, n_words, scan_tmp, st);
    hipLaunchKernelGGL(agx_k_special_emit, dim3(blocks), dim3(256), 0, st, *A, n_words, G);
}

}  // extern "C"
