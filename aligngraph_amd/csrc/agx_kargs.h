// agx_kargs.h — kernel argument blocks and launcher prototypes shared by agx_kernels.hip and agx_engine.cpp.
#pragma once
#include <hip/hip_runtime_api.h>
#include "agx_core.h"

struct agx_prep_args {
    const agx_hit *hits; const agx_run *runs; agx_dhit *dhit; agx_u32 n_hits, k, n_pos;
    agx_u32 *tile_cnt;        // [n_tiles] number of hits overlapping each tile
    agx_u32 *err;             // bit 0: same-strand mates; bit 1: alignment beyond the unit sequence
};

struct agx_bin_args { const agx_dhit *dhit; agx_u32 n_hits; const agx_u32 *tile_off; agx_u32 *cursor; agx_u32 *unsorted; };

struct agx_node_kargs {
    agx_sweep_args S;
    agx_u32 *pool_counter;     // next free node id
    agx_u32 *big_count; agx_u32 *big_list;   // tiles whose buckets did not fit in LDS
    agx_u32 *status;           // bit 0: node pool exhausted; bit 1: bucket overflow in the global-scratch pass
    const agx_u32 *tile_list;  // when non-null: the tiles to process (fallback pass), else all tiles
    agx_u32 n_list;
    agx_u32 *scratch;          // fallback pass: [n_list][AGX_NF*AGX_MAXV_BIG*64]
};

struct agx_edge_kargs { agx_sweep_args S; agx_edge_ovf *ovf; agx_u32 *ovf_count; agx_u32 ovf_cap; };

extern "C" {
void agx_launch_hit_prep(const agx_prep_args *, hipStream_t);
// exclusive scan of in[0..n] (n+1 entries, in[n] must be 0) into out[0..n]; out[n] = total
void agx_launch_exclusive_scan(const agx_u32 *in, agx_u32 *out, agx_u32 n, agx_u32 *tmp, hipStream_t);
void agx_launch_bin_fill(const agx_bin_args *, hipStream_t);
void agx_launch_tile_sort(const agx_u32 *tile_off, const agx_u32 *unsorted, agx_u32 *sorted, agx_u32 n_tiles, hipStream_t);
void agx_launch_node_sweep(const agx_node_kargs *, hipStream_t);
void agx_launch_node_sweep_big(const agx_node_kargs *, hipStream_t);
void agx_launch_edge_sweep(const agx_edge_kargs *, hipStream_t);
// walk preparation (agx_core.h): per-position side counts; then (after the scan) ids, records and overflow edges
void agx_launch_side_count(const agx_compact_args *, hipStream_t);
void agx_launch_compact(const agx_compact_args *, hipStream_t);
}
