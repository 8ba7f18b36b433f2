/* agx.h — C-ABI of libagx.so, the MI355X engine for AlignGraph's per-unit graph build + extend path.
 *
 * The reference (baoe/AlignGraph, one C++03 file "AG" = AlignGraph/AlignGraph.cpp) has no plugin or FFI
 * interface; the drop-in seam is the body of its unit loop (AG:4765-4783):
 *
 *     loadGenome(genome, u);  loadContigAlignment(genome, u);  loadReadAlignment(genome, k, iv, u, mrl);
 *     extendContigs(genome, coverage, k, u);  scaffoldContigs(genome, u);
 *
 * which reads tmp/_genome.u.fa, tmp/_contigs.fa, tmp/_contigs_genome.u.psl, tmp/_reads.fa and
 * tmp/_reads_genome.u.bowtie and writes tmp/_initial_contigs.u.fa, tmp/_pre_extended_contigs.u.fa and
 * tmp/_extended_contigs.u.fa.  agx_run_unit() replaces exactly those five calls (INTEGRATION.md shows the
 * three-line patch); the finer-grained entry points below split the same work at the packed-array boundary
 * so that a caller can keep inputs resident on the device and time host parsing, upload, kernels and the
 * walk separately.
 *
 * Plain C: pointers and sizes only, no C++ or torch types.  Every function returns AGX_OK (0) or a negative
 * AGX_E_* code; the message is available from agx_unit_error().  Nothing here ever exits the process or throws
 * across the boundary (the reference prints to stdout and exit(-1)s; INTEGRATION.md maps codes back to its
 * messages).  There is no CPU fallback: without a HIP device agx_unit_create() fails with AGX_E_NOGPU.
 *
 * Threading: an agx_unit is single-owner.  Distinct units may be driven concurrently, on one device or several.
 *
 * Device memory: ONE process per device is assumed.  The first unit of 8 GB or more makes a per-device region of AGX_REGION_PERCENT (default 85) per cent of the HBM
 * that is free at that moment and keeps it while anything lives in it (csrc/agx_mem.h: freed HBM stalls the next hipMalloc for seconds); a host that runs several
 * processes on one device sets the percentage per process, or AGX_NO_REGION=1 (every block then comes from hipMalloc or the cache of whole blocks).
 */
#ifndef AGX_H
#define AGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_OK 0
#define AGX_E_IO (-1)          /* "CANNOT OPEN FILE!"                                   AG:317, 356, 401, 849, 1274 */
#define AGX_E_FORMAT (-2)      /* "BROKEN BOWTIE FILE", "unknown character: c"           AG:1253, 267 */
#define AGX_E_UNSUPPORTED (-3) /* inputs the reference would index out of bounds on (DESIGN.md "Rejected inputs") */
#define AGX_E_ALIGNMENT (-4)   /* "BOWTIE ALIGNMENT ERROR" (mates on the same strand)    AG:1669 */
#define AGX_E_DEVICE (-5)      /* HIP runtime error */
#define AGX_E_ARG (-6)
#define AGX_E_OVERFLOW (-7)    /* more node variants at one position than the engine holds (AGX_MAXV_HUGE in agx_core.h: 1024; the reference's vector<KMer> is unbounded, AG:1375-1390) */
#define AGX_E_NOGPU (-8)

typedef struct agx_unit agx_unit;

/* Scalars of the unit loop: --kMer, --insertVariation, --coverage (AG:4701) and the compile-time BATCH (AG:37). */
typedef struct {
    uint32_t k;                /* default 5 */
    uint32_t insert_variation; /* default 50 */
    uint32_t coverage;         /* default 20 */
    uint32_t batch;            /* pairs per read batch; 0 = 1000000 (AG:37) */
    int32_t device;            /* HIP device ordinal */
    uint32_t flags;            /* AGX_FLAG_* */
} agx_params;

#define AGX_FLAG_KEEP_COUNTS 1u /* keep per-node coverage and base votes on the device for agx_unit_graph() */
#define AGX_FLAG_TIME_SECTIONS 4u /* time every section of a build (agx_stats ms_prep .. ms_compact); without it only ms_node_sweep is measured: an event
                                     record between two kernels costs the stream about as much as a small kernel */
#define AGX_FLAG_SPARSE_MIN  2u /* test hook: download node records of the side ids only, read all others one by one from the device */
#define AGX_FLAG_ONE_SHOT    8u /* the unit is uploaded ONCE after its inputs were handed over (the application's flow, agx_run_unit): once they are in HBM the
                                     staged input arrays are dead, and the download lands in that pinned memory instead of mapping and pinning 5 bytes per
                                     position of fresh memory per unit.  Another upload needs the inputs handed over again (agx_unit_load_files / push_pairs) */

/* ---- packed inputs -------------------------------------------------------------------------------- */

/* n read bases from read index q sit on reference offsets t, t+1, ...  (Segment, AG:44-49) */
typedef struct { uint32_t q, t, n; } agx_run;

/* One (pair, hit) that passed the identity filter of loadReadAli (AG:1261), in SAM order. */
typedef struct {
    uint32_t slot1;         /* mate1's slot in the read-base blob; mate2 is slot1+1 */
    uint32_t pos1, pos2;    /* "simple" mates (one M run covering the whole read): reference offset of read index 0 */
    uint32_t runs1, runs2;  /* first run in the run pool (non-simple mates) */
    uint16_t nruns1, nruns2; /* 0 = simple */
    uint16_t len;           /* read length; mates are equal length (AG:3454) */
    uint8_t rev1, rev2;     /* SAM FLAG 0x10 of each mate */
    uint8_t back;           /* number of earlier kept hits of the same pair */
    uint8_t pad[3];         /* ignored */
} agx_hit;

/* ContiMer (AG:51-62) inside a unit: the chromosome id is always 0 */
typedef struct { uint32_t cid, coff, next_off, next_item; char nuc; char pad[3]; } agx_contimer;

typedef struct {
    const agx_hit *hits; uint64_t n_hits;
    const agx_run *runs; uint64_t n_runs;
    const char *bases;      /* read slot s occupies bases[s*stride .. s*stride+len) in reads-file orientation */
    uint32_t stride, n_slots;
} agx_pair_batch;

/* ---- outputs --------------------------------------------------------------------------------------- */

typedef struct {
    char *initial_contigs; size_t initial_len;   /* bytes of tmp/_initial_contigs.u.fa      (AG:1179-1216) */
    char *pre_extended;    size_t pre_len;       /* bytes of tmp/_pre_extended_contigs.u.fa (AG:2176-2189) */
    char *extended;        size_t extended_len;  /* bytes of tmp/_extended_contigs.u.fa     (AG:2451-2463) */
} agx_result;

typedef struct {
    uint64_t n_pos, n_ref, n_hits, n_runs, n_nodes, n_tiles, n_tile_entries, n_big_tiles, n_edge_overflow;
    uint64_t pairs_in_file, sam_line_pairs;
    double ms_parse, ms_thread, ms_upload, ms_prep, ms_bin, ms_node_sweep, ms_node_big, ms_edge_sweep, ms_compact, ms_download, ms_walk;
    uint32_t node_sweep_launches, edge_sweep_launches;
    uint64_t n_walk_ids, n_special, n_fetched, download_bytes;   /* walk graph: ids, node records downloaded, records fetched one by one, D2H bytes */
    double ms_edge_fast, ms_edge_slow;                           /* the two kernels of ms_edge_sweep: pass A (lanes = positions), pass B (lanes = hits) */
    uint64_t n_mid_tiles;                                        /* tiles swept again with wider LDS buckets (n_big_tiles: of those, again with global scratch) */
    double ms_build_span;                                        /* with AGX_FLAG_TIME_SECTIONS: device time from the first to the last command of the build (all kernels and the gaps between them) */
    double ms_stage;                                             /* packing the handed-over arrays into pinned upload buffers (agx_unit_stage) */
    double ms_upload_dev;                                        /* with AGX_FLAG_TIME_SECTIONS: device time of the unit's upload copies on the device's upload stream */
    uint64_t upload_bytes, device_bytes;                         /* bytes copied host -> HBM by the upload; HBM held by the unit */
    uint64_t pinned_bytes_cached, device_bytes_cached;           /* free blocks in the library's pinned-host and device caches (agx_pool_trim releases them) */
    uint64_t n_spilled;                                          /* node ids taken from the pool's spill area (regions whose slice was full) */
    uint32_t build_attempts, from_cache;                         /* build_attempts: 1 unless a capacity had to grow and the build was repeated; from_cache: the unit was
                                                                    loaded from its cache file (agx_unit_cache_build), not from the text files */
    uint32_t dense_lists, rows_by_reference;                     /* dense_lists: 0 = every tile's hit list came out of its window of the hits' tile order (agx_k_tile_fill); 1 = some hits
                                                                    span more tiles than the window looks back over (long deletions) and came through the list of long hits; 2 = more than
                                                                    1024 such hits: every list of the unit was made by scatter (agx_k_bin_fill + agx_k_tile_sort).
                                                                    rows_by_reference: read rows the upload sent as their differences from the reference under their first hit's
                                                                    alignment (AGX_ROW_DIFF, engine: stage_rows; 0: all rows crossed as 2-bit classes) */
} agx_stats;

/* Node/edge tables in canonical numbering (position-major, variant order), for parity tests. malloc'd; free with agx_graph_free. */
typedef struct {
    uint32_t n_pos, n_nodes, n_edges;
    uint32_t *node_start;   /* [n_pos+1] */
    uint32_t *node_key;     /* [n_nodes*6] contigID, contigOffset, contigID0, contigOffset0, chromosomeID0, chromosomeOffset0 */
    int32_t *node_cnt;      /* [n_nodes*6] coverage, A, C, G, T, N   (needs AGX_FLAG_KEEP_COUNTS, else all -1) */
    uint32_t *node_slen;    /* [n_nodes] length of the stored k-mer string */
    uint32_t *edge_start;   /* [n_nodes+1] */
    uint32_t *edge_dst;     /* [n_edges] sorted per node */
} agx_graph;

/* ---- entry points ------------------------------------------------------------------------------------ */

const char *agx_version(void);
int agx_device_count(void);                                   /* HIP devices visible; 0 when there is no GPU */
int agx_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);   /* HBM of one device (hipMemGetInfo) */
int agx_selftest_scan(int device, uint32_t n, uint32_t seed);   /* test hook: the build's one-launch prefix scan over n pseudo-random counts against a host scan; AGX_OK or AGX_E_DEVICE */

int agx_unit_create(const agx_params *p, agx_unit **out);
void agx_unit_destroy(agx_unit *u);
const char *agx_unit_error(const agx_unit *u);

/* Packed-array boundary.  Pointers are borrowed for the call; the unit keeps its own copy. */
int agx_unit_set_reference(agx_unit *u, const char *bases, uint32_t n);                       /* replaces loadGenome, AG:287-320 */
int agx_unit_set_contig_threads(agx_unit *u, const char *appended, uint32_t n_appended,      /* result of updateGenomeWithContig, AG:884-1217 */
                                const uint32_t *cm_start, const agx_contimer *cm, uint32_t n_cm,
                                const char *initial_contigs, size_t initial_len);
int agx_unit_push_pairs(agx_unit *u, const agx_pair_batch *b);                                /* result of loadSeq + loadReadAli, AG:361-404, 1233-1277 */

/* Host loaders: the reference's own text files -> the three calls above. */
int agx_unit_load_files(agx_unit *u, const char *tmp_dir, int unit);

/* tmp/_reads.fa is the same file for every unit of a run, and the reference reads all of it for each unit (loadSeq under
 * loadReadAlignment, AG:1880).  An agx_reads maps it once and indexes its records; the _shared loaders then touch only the reads a
 * unit's own alignments name.  Immutable once opened: any number of host threads may load units from one agx_reads. */
typedef struct agx_reads agx_reads;
int agx_reads_open(const char *reads_fa, agx_reads **out, char *err, size_t err_len);
void agx_reads_close(agx_reads *reads);
int agx_unit_load_files_shared(agx_unit *u, const char *tmp_dir, int unit, const agx_reads *reads);   /* reads == NULL: same as agx_unit_load_files */

/* The binary cache of a unit's parsed inputs: tmp_dir/_agx_unit.<unit>.bin holds the unit's staged arrays, made from the
 * five text files once — AlignGraph_amd does it where the reference distributes the alignments (AG:3545-3579).  agx_unit_load_files* take
 * the cache instead of the text whenever it is current (same BATCH, the text files unchanged in size and modification time; AGX_NO_CACHE=1 in
 * the environment turns that off).  p: only batch and device matter. */
int agx_unit_cache_build(const agx_params *p, const char *tmp_dir, int unit, const agx_reads *reads, char *err, size_t err_len);
int agx_unit_cache_save(agx_unit *u, const char *tmp_dir, int unit);   /* the same file from a unit that agx_unit_load_files has just loaded from the text of (tmp_dir, unit) */

int agx_unit_stage(agx_unit *u);                 /* packs what was handed over into pinned upload buffers (read bases as 4-bit classes); done by agx_unit_load_files,
                                                    implied by agx_unit_upload after agx_unit_push_pairs */
int agx_unit_hbm_needed(agx_unit *u, uint64_t *bytes);   /* the HBM block agx_unit_upload will take for this (loaded) unit at its first-guess capacities: what a caller that shares a
                                                    device between units of very different sizes admits them by (AlignGraph_amd); a build that has to grow a capacity takes more */
int agx_unit_upload(agx_unit *u);                /* staged arrays -> HBM: one device block, asynchronous copies behind those of the device's earlier uploads; returns without waiting */
int agx_unit_build(agx_unit *u);                 /* kernels: updateGenomeWithRead/updateKMer (AG:1635-1870, 1353-1624) + filterLowCoverage (AG:1904-1918) */
int agx_unit_download(agx_unit *u);              /* HBM -> pinned host memory (the whole walk graph; returns when it is there).  Only needed by a caller that wants the unit's HBM back
                                                    before the walk (agx_unit_trim): agx_unit_finish downloads what has not been downloaded, and does it better */
int agx_unit_finish(agx_unit *u, agx_result *r); /* extdContigs1/2 + scaffoldContigs (AG:1954-2464) on the host.  On a unit that has not been downloaded (r06) the download is STREAMED: the
                                                    walk graph comes down in position windows from the front and the walk begins on what has landed — the walkers of a large unit wait for
                                                    their windows, the first one for all of them (csrc/agx_engine.cpp: begin_streamed_download; AGX_NO_STREAM_DOWNLOAD=1: the whole download first) */
void agx_result_free(agx_result *r);               /* (the buffers are malloc'd; the library keeps up to 16 GB of the ones given back here for the outputs of the next units — fresh memory for them is a
                                                    sixth of a whole-human job's host CPU time —, agx_pool_trim(-1) frees them) */
int agx_unit_trim(agx_unit *u, uint64_t *freed);  /* after agx_unit_download: gives the part of the unit's HBM that the host walk cannot ask for (three quarters of it) back to the device's memory
                                                    region, so that the next unit is admitted when this one's DOWNLOAD is done, not when its walk is; *freed = bytes given back (0: nothing to give —
                                                    small units on a device without a region).  The unit is no longer built afterwards: agx_unit_finish still works, another build uploads again */
int agx_unit_release(agx_unit *u);               /* gives the unit's HBM and download buffers back to the library's caches; the staged inputs stay: upload again = a new unit (not AGX_FLAG_ONE_SHOT units: their inputs are gone after the download) */
void agx_pool_trim(int device);                  /* device >= 0: frees the cached HBM blocks of that device; -1: frees the cached (and retired) pinned host blocks;
                                                    -2: retires the cached pinned host blocks (never handed out again, unmapped by the next -1): what a
                                                    measurement loop uses so that every job maps and registers fresh buffers without paying for unmapping old ones */

int agx_unit_stats(const agx_unit *u, agx_stats *s);
int agx_unit_graph(agx_unit *u, agx_graph *g);   /* after agx_unit_build */
void agx_graph_free(agx_graph *g);

/* The five-call seam in one call.  write_files != 0 also writes the three files under tmp_dir like the reference does. */
int agx_run_unit(const agx_params *p, const char *tmp_dir, int unit, int write_files, agx_result *r, char *err, size_t err_len);
int agx_run_unit_shared(const agx_params *p, const char *tmp_dir, int unit, int write_files, const agx_reads *reads, agx_result *r, char *err, size_t err_len);

#ifdef __cplusplus
}
#endif
#endif /* AGX_H */
