// agx_kargs.h — kernel argument blocks and launcher prototypes shared by agx_kernels.hip and agx_engine.cpp.
#pragma once
#include <hip/hip_runtime_api.h>
#include "agx_core.h"

struct agx_prep_args {
    const agx_whit *whits; const agx_wside *sides;      // [n_hits] wire records in SAM file order: a hit's number is its place in the file, the order the tile lists are sorted into
    const agx_run *runs; agx_dhit *dhit; agx_u32 n_hits, k, n_pos;      // dhit[i]: the derived record of the i-th hit IN TILE ORDER (hit perm[i])
    agx_u32 *tile_cnt;        // [n_tiles] number of hits overlapping each tile
    agx_u32 *err;             // bit 0: same-strand mates; bit 1: alignment beyond the unit sequence; bit 2: the order the staging made is not the order of the hits' first tiles
    // r05: the hits are walked in the order of their first tile (perm: stage_order; tile_first[t] = hits in front of tile t's own in that order), so a tile's list is a
    // filter over a window of the order (agx_k_tile_fill).  ckey[i] = the last tile hit i reaches (NONE: dropped, or LONG — it spans `lookback` tiles or more and is
    // listed in long_list instead; more than AGX_LONG_MAX of those: every list of the unit is made by scatter, agx_k_bin_fill + agx_k_tile_sort)
    const agx_u32 *perm, *tile_first; agx_u32 *ckey; agx_u32 lookback; agx_u32 *long_list, *long_count;
    // r06, tile-ordered upload (tiled != 0): whits[i] IS the i-th hit of the tile order — its `row` field carries the hit's number, its left-mate row is row i, AGX_WF_DUP says
    // what the file neighbours would have said — and perm_out[i] = that number is written here for the tile lists (4 bytes per hit less to upload, no gather of wire records)
    agx_u32 tiled; agx_u32 *perm_out;
};

// the fallback (more than AGX_LONG_MAX long hits): every hit takes its places in dense lists from a counter per tile
struct agx_bin_args { const agx_dhit *dhit; agx_u32 n_hits; const agx_u32 *tile_off; agx_u32 *cursor; agx_u32 *unsorted; agx_u32 cap;   // cap: entries the lists can hold
                      const agx_u32 *long_count; };

struct agx_node_kargs {
    agx_sweep_args S;
    agx_u32 tile_lo, tile_hi;  // pass 0 sweeps tiles [tile_lo, tile_hi): a unit's first build sweeps a window of tiles as soon as its read rows have landed (r06)
    agx_u32 *pool_cnt;         // node ids handed out per region (counter r at pool_cnt[r * AGX_REGION_PAD]); keeps counting past the slice's end
    const agx_u32 *region_off; // [regions + 1] first node id of every region's slice of the pool
    agx_u32 spill_lo; agx_u32 *spill_cnt;   // ids [spill_lo, pool_cap) behind the slices, for regions whose slice is full (one counter)
    agx_u32 *mid_count; agx_u32 *mid_list;   // tiles whose buckets did not fit pass 0's
    agx_u32 *big_count; agx_u32 *big_list;   // tiles whose buckets did not fit pass 1's either
    agx_u32 fallback_queued;   // the two fallback passes are queued behind the main pass (a unit's builds start without them: most units never overflow a bucket)
    agx_u32 *status;           // bit 3: a tile overflowed the main pass and no fallback pass is queued; bit 0: a region's slice of the node pool exhausted; bit 1: bucket overflow in the global-scratch pass; bit 2: tile lists too small;
                               // bit 4 (set by agx_k_tile_fill, read by the main pass): the tile lists were not made
    agx_u32 list_cap;          // capacity of the tile lists: a tile whose list ends beyond it is skipped (the host re-runs with larger lists)
    const agx_u32 *big_n;      // passes 1 and 2: number of tiles in mid_list / big_list, read on the device (no host round trip)
    const agx_u32 *mid_n;
    agx_u32 *scratch;          // fallback pass: one [AGX_NF*AGX_MAXV_BIG*64] bucket area per resident wavefront
    agx_u32 *huge_count; agx_u32 *huge_list; const agx_u32 *huge_n; agx_u32 *scratch_huge; agx_u32 huge_queued;   // pass 3 (AGX_MAXV_HUGE variants): tiles pass 2 gave up on
    agx_u32 *slow_list; agx_u32 *slow_count;   // the edge build's pass-B list: the sweep itself enters multi-variant positions with a position-skipping step
};

struct agx_edge_kargs {
    agx_sweep_args S; agx_edge_ovf *ovf; agx_u32 *ovf_count; agx_u32 ovf_cap; agx_u32 list_cap;
    agx_u32 *slow_list; agx_u32 *slow_count;   // positions that need the per-hit pass (device-side list)
    agx_u32 n_hits; const agx_u32 *jump_list; agx_u32 n_jump;      // pass J: the hits whose left mate has several runs (listed by the host's staging)
    const agx_u32 *abort;                      // the node sweeps' status word: non-zero = the node table is incomplete, do nothing
    const agx_u32 *big_list; const agx_u32 *big_n;   // tiles the fallback pass wrote (their edges are all pass A/B's)
};
#define AGX_SLOW_WAVES 8192u    // wavefronts of the per-hit edge pass if the occupancy query fails (normally: as many as are resident at once)

extern "C" {
// one kernel that zeroes up to eight u32 ranges
struct agx_zero_args { agx_u32 *p[8]; agx_u32 n[8]; };
void agx_launch_zero(const agx_zero_args *, hipStream_t);
void agx_launch_cm_tables(const void *cnt_runs, const void *cnt_chunks, agx_u32 n_cnt_chunks, const agx_cmseg *segs, const void *seg_chunks, agx_u32 n_seg_chunks,
                           agx_u32 *cm_start, agx_cmkey *cm, agx_cmhead *head, agx_u32 n_pos, agx_u32 n_cm, hipStream_t);      // cm_start[n_pos + 1], cm, n_pos + 1 heads from the count runs and the conti-mer runs (agx_core.h: agx_cntrun, agx_chunk)
void agx_launch_expand_codes(const void *packed, void *vcodes, size_t n_bases16, const unsigned long long *other, size_t n_other, hipStream_t);      // 2-bit base classes (agx_pack_classes2) + the listed other bases -> agx_vote_code bytes; n_bases16 a multiple of 16
// the rows from their upload form (differences against the reference under each row's anchor hit, or the 2-bit classes: agx_core.h) -> agx_vote_code bytes, then the listed other bases.
// cnt: zeros up to the next multiple of 64 rows; anchor_bits: eight words readable behind any block's first anchor; wref: the packed reference, one 32-bit word of slack behind
// position n_pos + 15; stride <= AGX_ROW_MAXSTRIDE
void agx_launch_expand_rows(const void *whits, agx_u32 nh, const void *wsides, const void *wruns, const agx_u32 *anchor_bits, const agx_u32 *block_first, const agx_u8 *cnt, const agx_u32 *block_off,
                            const agx_u16 *units, const void *wref, void *vcodes, agx_u32 n_rows, agx_u32 stride, const unsigned long long *other, size_t n_other, hipStream_t);
void agx_launch_patch_codes(const unsigned long long *other, size_t n_other, void *vcodes, hipStream_t);      // the listed bases alone (indices into the whole vote-code array)
void agx_launch_expand_runs(const void *wruns, agx_run *runs, agx_u32 n_runs, hipStream_t);      // wire formats (agx_core.h) -> working arrays
void agx_launch_expand_ref(const void *packed, void *ref, size_t n_pos16, const void *refx, agx_u32 n_refx, hipStream_t);      // 2-bit reference bases + the stretches of other bytes -> letters; n_pos16 a multiple of 16
void agx_launch_hit_prep(const agx_prep_args *, hipStream_t);
// exclusive scan of in[0..n] (n+1 entries, in[n] must be 0) into out[0..n]; out[n] = total
void agx_launch_exclusive_scan(const agx_u32 *in, agx_u32 *out, agx_u32 n, agx_u32 *tmp, hipStream_t);
void agx_launch_exclusive_scan1(const agx_u32 *in, agx_u32 *out, agx_u32 n, unsigned long long *desc, hipStream_t);      // one launch; desc: ceil((n+1)/4096) zeroed words
void agx_launch_bin_fill(const agx_bin_args *, hipStream_t);
// the tile lists as the sweeps read them (32-byte records in SAM order): agx_k_tile_fill from the window of the tile order, or — the fallback — agx_k_tile_sort from bin_fill's dense lists
struct agx_fill_args { const agx_u32 *tile_off, *tile_first, *perm, *ckey; const agx_dhit *dhit; const agx_run *runs; void *recs; agx_u32 *scratch /* = bin_fill's `unsorted` */;
                       agx_u32 n_tiles, cap, k, lookback; const agx_u32 *long_list, *long_count; agx_u32 *err;
                       agx_u32 *status; agx_u32 dense_queued; };      // status bit 4 (16): this unit needs the scatter fallback and it is not queued (dense_queued = 0): the build is repeated with it
void agx_launch_tile_fill(const agx_fill_args *, hipStream_t);
void agx_launch_tile_sort(const agx_fill_args *, hipStream_t);
void agx_launch_node_sweep(const agx_node_kargs *, hipStream_t);
void agx_launch_node_sweep_big(const agx_node_kargs *, hipStream_t);
void agx_launch_edge_sweep(const agx_edge_kargs *, hipStream_t);                   // pass A (lanes = positions)
void agx_launch_edge_jump(const agx_edge_kargs *, hipStream_t);    // pass J (lanes = hits: steps that skip positions)
void agx_launch_edge_slow(const agx_edge_kargs *, hipStream_t);           // pass B (lanes = hits of the slow positions pass A listed)
// walk preparation (agx_core.h): after the scan of the side counts the node sweep left behind: ids, records and overflow edges
// n_nodes / n_ovf are read from device memory (the node-pool and overflow counters), so no host round trip separates the sweeps
// from the walk preparation; the grids are sized by the capacities.
void agx_launch_copy_out(void *const *dst, const void *const *src, const size_t *bytes, int n, hipStream_t);      // HBM -> registered host memory, by a kernel
void agx_launch_fetch_records(const agx_compact_args *, agx_u32 first, agx_u32 stride, agx_u32 rows, agx_u32 width, agx_walknode *out, hipStream_t);
void agx_launch_compact(const agx_compact_args *, const agx_u32 *chain_end, agx_u32 n_chain_end, const agx_u32 *n_ovf_dev, agx_u32 ovf_cap, hipStream_t);
// special ids: bitmap, rank scan (desc: the one-launch scan; null: the three-launch one with scan_tmp), records; block 0 of the last kernel also leaves the
// totals the host reads: out[0..2] = *a, *b, sp_rank[n_words]; *sum = nodes handed out = the sum of the region counters
// r06: where a streamed download cuts the walk graph (agx_engine.cpp: begin_streamed_download): piece w holds main ids [64 * word[w], 64 * word[w + 1]) (the last one: up to n_pos)
// and the side ids of those positions.  The host needs, before it can queue the copies, the special ids in front of every cut (sparse-table ranks: sp_rank at the cut's word)
// and the side ids in front of it (tile_side_start at the cut's tile): block 0 of the build's last kernel leaves them next to the counters, cut_out[0 .. n] = ranks (entry n: all
// main ids), cut_out[AGX_DL_PIECES + 1 ..] = side ids.
#define AGX_DL_PIECES 16u
struct agx_cut_args { agx_u32 n; agx_u32 word[AGX_DL_PIECES + 1]; agx_u32 *cut_out; };
void agx_launch_special(const agx_compact_args *, agx_u32 n_words, agx_u32 *sp_rank, agx_u32 *scan_tmp, unsigned long long *desc,
                        agx_u32 *out, const agx_u32 *a, const agx_u32 *b, const agx_u32 *pool_cnt, agx_u32 regions, agx_u32 *sum, const agx_cut_args *cuts, hipStream_t);
#define AGX_MID_WAVES 3072u     // wavefronts of pass 1 (3 per SIMD fit its LDS buckets); they stride over the list of tiles pass 0 gave up on
#define AGX_BIG_WAVES 256u      // resident wavefronts of the global-scratch fallback pass
#define AGX_HUGE_WAVES 32u      // wavefronts of pass 3 (0.85 MB of scratch each)
void agx_launch_node_sweep_huge(const agx_node_kargs *, hipStream_t);
}
