// agx_core.h — data layout and per-lane algorithm of the MI355X graph-build engine.
//
// Everything here is written once and compiled twice: by hipcc into the gfx950 kernels
// (agx_kernels.hip) and by g++ into the test-only serial executor (tests/hostsim) that lets the
// CPU test-suite check the re-formulated algorithm against the oracle without a GPU.
// The product library never runs these functions on the host.
//
// Re-formulation of the reference's node build (AG = /root/reference/AlignGraph/AlignGraph.cpp):
//
//  * The reference applies one "event" per aligned read index (updateGenomeWithRead, AG:1635-1870 ->
//    updateKMer, AG:1353-1624): a k1 half at position P (match-or-insert, coverage++, base vote) and a
//    k2 half at the next position N (match-or-insert only).  The k2 half of an event and the k1 half of
//    the same read's next event address the same position with the same key, so here every
//    (hit, position) pair carries ONE "arrival": K1 (count + vote), CHAIN (count, no vote, empty k-mer —
//    the positions the reference walks through when a read insertion sits next to a reference gap,
//    AG:1730-1751) or K2ONLY (the last position a read reaches, AG:1484-1500: insert with coverage 0).
//  * A bucket's variants are only ever appended, and `compatible` (AG:1293-1312) is reflexive, so the
//    variant an arrival resolves to is the first compatible variant of the FINAL bucket.  Node build is
//    therefore one in-order sweep per position (lanes = positions, hits applied in SAM order).
//  * An event's edge (AG:1589-1623) joins the variants its k1 half touched at P with the variants its k2 half
//    touched at N — i.e. with the variants the same hit's arrival touched at N.  Where N = P+1 lies in the same
//    tile the two lanes exchange those variant sets while they sweep and the edges are written with the nodes;
//    everything else (tile boundaries, steps that skip positions, overflowed buckets) is re-resolved against the
//    final buckets by the edge passes.  Edge sets are sets: no dependence on scheduling, bit-exact by construction.
#pragma once
#include <stdint.h>
#include "../../include/agx.h"

#if defined(__HIPCC__)
#define AGX_HD __host__ __device__ __forceinline__
#else
#define AGX_HD inline
#endif

typedef uint32_t agx_u32;
typedef uint16_t agx_u16;
typedef uint8_t agx_u8;

#define AGX_NONE 0xFFFFFFFFu
#define AGX_TILE 64u          // positions per tile = lanes per wavefront
// The node pool is cut into one slice per AGX_REGION_TILES consecutive tiles, each with its own allocation counter (agx_k_node_sweep).
#ifndef AGX_REGION_TILES
#define AGX_REGION_TILES 32u
#endif
#define AGX_REGION_PAD 32u      // counters sit 128 bytes apart
// The node sweep runs in three passes with growing buckets; a tile whose bucket overflows is swept again by the next pass.
#ifndef AGX_MAXV_LDS
#define AGX_MAXV_LDS 2u       // pass 0, every tile: variants per position held in LDS (2: 6.5 KB per wavefront -> 6 wavefronts per SIMD; the sweep
                              // waits on memory more than it computes: 0.66 ms with 2, 0.73 ms with 3 (4 wavefronts per SIMD) on the bench unit)
#endif
#ifndef AGX_SWEEP_FIRST
#define AGX_SWEEP_FIRST 1        // straight-line store of a position's first variant (0: everything that leaves the fast path takes the general path)
#endif
#define AGX_MAXV_MID 4u       // pass 1, LDS again (13 KB per wavefront): the widest bucket whose x -> x+1 edges still fit the sweep's edge matrix
#define AGX_MAXV_BIG 64u      // pass 2: buckets in global scratch
#define AGX_MAXV_HUGE 1024u   // pass 3, queued only for a unit that has met a position beyond 64 variants (r04: 1024, node_cnt is 16 bits wide; r03: 255, one byte) can count
#define AGX_MAXE 4u           // out-edges stored inline per node; more go to the overflow list
#define AGX_EP25 25           // 5*EP (AG:39, 1296)

// ---- packed inputs (host -> device) -----------------------------------------------------------------

// agx_run and agx_hit are part of the public packed-array boundary: include/agx.h

// ---- wire formats: what crosses PCIe ------------------------------------------------------------------------------------
// A unit's upload is what a cfg3 job waits for (1.5 GB at 56 GB/s = 29 of its 50 ms in r02), so the arrays cross in a packed form and a kernel
// at the head of the unit's first build expands them into the working forms above (agx_k_expand_hits, agx_k_expand_ref): hits 32 -> 16 bytes
// (+ 12 for the three in eight that have a multi-run mate), runs 12 -> 8, reference bases 8 -> 2 bits.
enum { AGX_WF_REV1 = 1, AGX_WF_REV2 = 2, AGX_WF_LEFT2 = 4, AGX_WF_RUNS1 = 8, AGX_WF_RUNS2 = 16,
       AGX_WF_DUP = 32 };      // (tile-ordered upload, r06: the rule of AG:1650-1655 — a later hit of a pair that lands on an earlier one is dropped — decided where the arrays are staged: the hit's file neighbours are not its neighbours in that order)
struct agx_whit {             // a: reference offset of mate1's read index 0 if mate1 is one full-length run, b: the same for mate2.  A hit with a multi-run
    agx_u32 a, b, row;        // mate has a side record; its index sits in the field of the first mate that has runs (a if RUNS1, else b)
    agx_u16 len; agx_u8 flags, back;
};
struct agx_wside { agx_u32 runs1, runs2, nruns; };      // first run of each mate in the run pool; nruns = nruns1 | nruns2 << 16
struct agx_wrun { agx_u32 t; agx_u16 q, n; };           // read lengths stay below 65536 (the loaders refuse longer reads)
AGX_HD agx_whit agx_pack_hit(const agx_hit &h, agx_u32 side_index) {      // h: a staged hit (slot1 = row, pad[0] = left mate)
    agx_whit w; w.a = h.nruns1 ? 0u : h.pos1; w.b = h.nruns2 ? 0u : h.pos2; w.row = h.slot1; w.len = h.len; w.back = h.back;
    w.flags = (agx_u8)((h.rev1 ? AGX_WF_REV1 : 0) | (h.rev2 ? AGX_WF_REV2 : 0) | ((h.pad[0] & 1u) ? AGX_WF_LEFT2 : 0) | (h.nruns1 ? AGX_WF_RUNS1 : 0) | (h.nruns2 ? AGX_WF_RUNS2 : 0));
    if (h.nruns1) w.a = side_index; else if (h.nruns2) w.b = side_index;
    return w;
}
AGX_HD agx_hit agx_unpack_hit(const agx_whit &w, const agx_wside *sides) {
    agx_hit h; h.slot1 = w.row; h.len = w.len; h.back = w.back; h.rev1 = (w.flags & AGX_WF_REV1) ? 1 : 0; h.rev2 = (w.flags & AGX_WF_REV2) ? 1 : 0;
    h.pad[0] = (w.flags & AGX_WF_LEFT2) ? 1 : 0; h.pad[1] = h.pad[2] = 0;
    h.pos1 = (w.flags & AGX_WF_RUNS1) ? 0u : w.a; h.pos2 = (w.flags & AGX_WF_RUNS2) ? 0u : w.b; h.runs1 = h.runs2 = 0; h.nruns1 = h.nruns2 = 0;
    if (w.flags & (AGX_WF_RUNS1 | AGX_WF_RUNS2)) {
        const agx_wside sd = sides[(w.flags & AGX_WF_RUNS1) ? w.a : w.b];
        if (w.flags & AGX_WF_RUNS1) { h.runs1 = sd.runs1; h.nruns1 = (agx_u16)(sd.nruns & 0xFFFFu); }
        if (w.flags & AGX_WF_RUNS2) { h.runs2 = sd.runs2; h.nruns2 = (agx_u16)(sd.nruns >> 16); }
    }
    return h;
}
// reference bases: 2 bits each (A, C, G, T = 0..3, four per byte, position x in bits 2 * (x & 3) of byte x >> 2) + the stretches that are anything
// else (N runs, lower case), as runs of one byte value
struct agx_refx { agx_u32 pos, len, byte; };
AGX_HD agx_u32 agx_ref_code(agx_u32 c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u; }
AGX_HD char agx_ref_base(agx_u32 code) { return (char)((0x54474341u >> (8u * (code & 3u))) & 0xFFu); }      // "ACGT"

// per-position conti-mer key (the part of ContiMer, AG:51-62, that node build reads)
struct agx_cmkey { agx_u32 cid, coff; };
// per-position head of the conti-mer table, built on the device at upload: the first conti-mer of the position (NONE/NONE if there is
// none), how many there are and where they start.  One 16-byte load tells the node sweep everything the common case needs about a
// position instead of two offsets and a dependent key load.
struct agx_cmhead { agx_u32 cid, coff, n, start; };

// ---- derived per-hit record (device, written by hit_prep) ---------------------------------------------

enum { AGX_HF_AREV = 1, AGX_HF_SKIP = 2, AGX_HF_BNONE = 4 };      // BNONE (tile records only): the b mate is unaligned over the whole piece

struct agx_dhit {             // 40 bytes = five 8-byte words (agx_k_tile_fill reads it that way; padded to 48 bytes for 16-byte loads it was slower: HISTORY.md r05)
    agx_u32 a_t0, b_t0;       // simple mate: reference offset of read index 0 ("a" = the left mate, AG:1672-1679)
    agx_u32 a_runs, b_runs;
    agx_u32 a_slot;           // read slot of the a mate
    agx_u16 len, jstar;       // jstar: read index of the K2ONLY arrival, 0xFFFF if none
    agx_u16 a_nruns, b_nruns; // 0 = simple
    agx_u32 flags;
    agx_u32 x_lo, x_hi;       // first / last position that receives an arrival (x_lo > x_hi: none)
};

// ---- node table (device -> host) ----------------------------------------------------------------------

enum { AGX_NF_DEAD = 1, AGX_NF_CONTIG = 2, AGX_NF_EOVF = 4 };   // pruned (AG:1912-1915); contigOffset != -1 (AG:2004); edges in overflow list

// k-mer string reference: the k-mer of a node is reads[slot] (oriented) [q, q+len)
struct agx_sref { agx_u32 slot; agx_u32 qlen; };   // qlen = first | len<<16 | rev<<31; `first` is the stored (file-orientation) index of
                                                   // the k-mer's first base: forward reads run first, first+1, ..; reverse reads run first, first-1, .. complemented

// bucket fields while a tile is being swept (struct-of-arrays: field-major, then variant, then lane)
enum { AGX_F_CID = 0, AGX_F_COFF, AGX_F_CID0, AGX_F_COFF0, AGX_F_OFF0, AGX_F_COV, AGX_F_A, AGX_F_C, AGX_F_G, AGX_F_T, AGX_F_N, AGX_F_S0, AGX_F_S1, AGX_NF };
static_assert(AGX_F_C == AGX_F_A + 1 && AGX_F_G == AGX_F_A + 2 && AGX_F_T == AGX_F_A + 3 && AGX_F_N == AGX_F_A + 4, "agx_class_vote_code relies on the order of the vote counters");

struct agx_key { agx_u32 cid, coff, cid0, coff0, off0; };   // chromosomeID0 is 0 iff off0 != NONE inside a unit

AGX_HD int agx_absdiff(agx_u32 a, agx_u32 b) { int d = (int)(a - b); return d < 0 ? -d : d; }   // abs((int)(a-b)), AG:1296

// clauses of `compatible` (AG:1293-1312, OPTIMIZATION on): A on (contigID, contigOffset), B on the mate's, C on the mate position.
// Written as integer arithmetic with `&` / `|` (no short-circuit): on wave64 every lane-varying `&&` turns into an exec-mask
// branch, and these run once per (hit, position).  |a-b| <= win  <=>  (u32)(a - b + win) <= 2*win  (win >= 0, |a-b| < 2^31).
AGX_HD agx_u32 agx_within(agx_u32 a, agx_u32 b, int win) { return (agx_u32)(a - b + (agx_u32)win) <= 2u * (agx_u32)win ? 1u : 0u; }
// true unless both ids are set and equal and the offsets are further apart than win
AGX_HD agx_u32 agx_clause_ab(agx_u32 ac, agx_u32 ao, agx_u32 bc, agx_u32 bo, int win) {
    return ((agx_u32)(ac != bc) | (agx_u32)(ac == AGX_NONE) | agx_within(ao, bo, win)) & 1u;
}
AGX_HD agx_u32 agx_clause_c(agx_u32 ao, agx_u32 bo, int win) { return ((agx_u32)(ao == AGX_NONE) | (agx_u32)(bo == AGX_NONE) | agx_within(ao, bo, win)) & 1u; }

// Wave-uniform read of a table that the running kernel never writes.  On the device the load goes through the constant address space,
// so a uniform index makes it a scalar load (s_load_*, scalar cache) instead of 64 identical per-lane requests; the host reads normally.
#if defined(__HIP_DEVICE_COMPILE__)
AGX_HD agx_u32 agx_uload(const agx_u32 *p, size_t i) { return ((const __attribute__((address_space(4))) agx_u32 *)p)[i]; }
#else
AGX_HD agx_u32 agx_uload(const agx_u32 *p, size_t i) { return p[i]; }
#endif

// A bucket view: base points at this lane's column; element (variant v, field f) is base[(v*AGX_NF+f)*stride]
// packed (r06, pass 0 of the device's node sweep only): the six counters of a variant — coverage and the five votes — as 16-bit halves of three words, a variant = AGX_NFP words instead
// of AGX_NF: 5 KB of LDS per wavefront instead of 6.5, eight wavefronts per SIMD instead of six.  A list of more than 65 535 entries (no counter can pass 16 bits below that)
// leaves the tile to the next pass, whose buckets are not packed.  Counters are only reached through agx_cnt_*; agx_b serves the other fields in both layouts.
struct agx_bucket { agx_u32 *base; agx_u32 stride; agx_u32 maxv; agx_u32 packed; };
#define AGX_NFP 10u
AGX_HD agx_u32 &agx_b(const agx_bucket &b, agx_u32 v, agx_u32 f) { return b.base[(b.packed ? v * AGX_NFP + (f >= (agx_u32)AGX_F_S0 ? f - 3u : f) : v * AGX_NF + f) * b.stride]; }
AGX_HD agx_u32 &agx_cnt_word(const agx_bucket &b, agx_u32 v, agx_u32 f) { return b.base[(v * AGX_NFP + (agx_u32)AGX_F_COV + ((f - (agx_u32)AGX_F_COV) >> 1)) * b.stride]; }      // (packed layout: the word that holds counter f)
// counter update of a bucket word.  In the LDS node sweep it is a wave-private LDS add (one ds_add_u32 instead of
// read / wait / add / write); everywhere else it is the plain read-modify-write.
template <bool LDS_ADD> AGX_HD void agx_bucket_add(agx_u32 &ref, agx_u32 val) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (LDS_ADD) { (void)__hip_atomic_fetch_add(&ref, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); return; }
#endif
    ref += val;
}
AGX_HD agx_u32 agx_cnt_get(const agx_bucket &b, agx_u32 v, agx_u32 f) {
    if (b.packed) return (agx_cnt_word(b, v, f) >> (16u * ((f - (agx_u32)AGX_F_COV) & 1u))) & 0xFFFFu;
    return agx_b(b, v, f);
}
template <bool LDS_ADD> AGX_HD void agx_cnt_add(const agx_bucket &b, agx_u32 v, agx_u32 f, agx_u32 val) {
    if (b.packed) agx_bucket_add<LDS_ADD>(agx_cnt_word(b, v, f), val << (16u * ((f - (agx_u32)AGX_F_COV) & 1u)));
    else agx_bucket_add<LDS_ADD>(agx_b(b, v, f), val);
}
// a new variant's counters: coverage cov, no votes
AGX_HD void agx_cnt_init(const agx_bucket &b, agx_u32 v, agx_u32 cov) {
    if (b.packed) { agx_cnt_word(b, v, AGX_F_COV) = cov; agx_cnt_word(b, v, AGX_F_C) = 0; agx_cnt_word(b, v, AGX_F_T) = 0; }
    else { agx_b(b, v, AGX_F_COV) = cov; agx_b(b, v, AGX_F_A) = 0; agx_b(b, v, AGX_F_C) = 0; agx_b(b, v, AGX_F_G) = 0; agx_b(b, v, AGX_F_T) = 0; agx_b(b, v, AGX_F_N) = 0; }
}

AGX_HD bool agx_compatible(const agx_key &k, const agx_bucket &b, agx_u32 v, int iv) {
    const agx_u32 c = agx_b(b, v, AGX_F_CID), o = agx_b(b, v, AGX_F_COFF), c0 = agx_b(b, v, AGX_F_CID0), o0 = agx_b(b, v, AGX_F_COFF0), m = agx_b(b, v, AGX_F_OFF0);
    return (agx_clause_ab(k.cid, k.coff, c, o, AGX_EP25) & agx_clause_ab(k.cid0, k.coff0, c0, o0, 2 * iv + AGX_EP25) & agx_clause_c(k.off0, m, 2 * iv + AGX_EP25)) != 0;
}

// ---- read geometry ---------------------------------------------------------------------------------------

// reference offset of read index q of the b mate, or NONE.  Select-only: the loop count is wave-uniform, q is per lane.
AGX_HD agx_u32 agx_pos_b(const agx_dhit &d, const agx_run *runs, agx_u32 q) {
    if (d.b_nruns == 0) return d.b_t0 + q;
    agx_u32 res = AGX_NONE;
    for (agx_u32 i = 0; i < d.b_nruns; i++) { const agx_run r = runs[d.b_runs + i]; const agx_u32 o = q - r.q; res = (q >= r.q && o < r.n) ? r.t + o : res; }
    return res;
}

enum { AGX_AT_K1 = 0, AGX_AT_K2ONLY = 1, AGX_AT_CHAIN = 2 };

struct agx_arrival {
    agx_u32 has;              // 0: the hit contributes nothing at this position
    agx_u32 type, q, slen;    // q: read index whose base votes / starts the k-mer string
    agx_u32 p0;               // mate position (or NONE)
    agx_u32 has_succ, xs, p0s;  // successor arrival of the same hit: position and its mate position (the event's N / N0)
};

// What hit d contributes at position X (if anything).  Follows the event loop of AG:1681-1859; k = k-mer length.
//
// Written without per-lane branches: every `if` on a lane-varying condition costs exec-mask bookkeeping on wave64, and this runs
// once per (hit, position).  All loops below have wave-uniform trip counts (run counts of the hit), all decisions are selects.
AGX_HD agx_arrival agx_decode_arrival(const agx_dhit &d, const agx_run *runs, agx_u32 X, agx_u32 k) {
    agx_arrival a;
    const agx_u32 L = d.len, js = d.jstar;
    if (d.a_nruns == 0 && d.b_nruns == 0) {
        // A LINEAR PIECE: read indices qs .. qe (a_runs = qs | qe << 16) sit on a_t0 + q, their mate positions on b_t0 + q (or nowhere:
        // AGX_HF_BNONE), every index below jstar is an event source whose successor is the next position, jstar itself only receives the
        // k2 half (K2ONLY).  A hit whose mates are both one full-length run is one piece (qs = 0, qe = jstar: hit_prep writes it so);
        // agx_tile_record() cuts the other hits into the piece that covers a tile wherever there is one.
        // Hits with L <= k carry AGX_HF_SKIP and are never listed, so jstar is valid.
        const agx_u32 q = X - d.a_t0;                       // wraps to a huge value left of the piece
        const agx_u32 qs = d.a_runs & 0xFFFFu, qe = d.a_runs >> 16;
        const bool last = q == js, bnone = (d.flags & AGX_HF_BNONE) != 0;
        a.has = (q - qs) <= (qe - qs) ? 1u : 0u; a.type = last ? AGX_AT_K2ONLY : AGX_AT_K1; a.q = q;
        a.slen = last ? ((L - q) < k ? (L - q) : k) : k;
        a.p0 = bnone ? AGX_NONE : d.b_t0 + q; a.has_succ = last ? 0u : 1u; a.xs = X + 1; a.p0s = bnone ? AGX_NONE : d.b_t0 + q + 1;
        return a;
    }
    const agx_u32 lim = L > k ? L - k : 0u;
    const agx_u32 nr = d.a_nruns == 0 ? 1u : d.a_nruns;
    a.has = 0; a.type = AGX_AT_K1; a.q = 0; a.slen = 0; a.p0 = AGX_NONE; a.has_succ = 0; a.xs = X + 1; a.p0s = AGX_NONE;
    agx_u32 qsucc = AGX_NONE;                              // read index that supplies the successor's mate position (NONE: none)
    agx_run r = d.a_nruns == 0 ? agx_run{0u, d.a_t0, L} : runs[d.a_runs];
    for (agx_u32 i = 0; i < nr; i++) {
        const bool has_nx = i + 1 < nr;
        const agx_run nx = has_nx ? runs[d.a_runs + i + 1] : agx_run{0u, 0u, 0u};
        // X inside this run: index q.  Sources are the aligned indices below jstar; jstar is the first aligned index that is not a
        // source (at or beyond L-k, or the read's last aligned index) and only receives the k2 half (AG:1484-1500).
        const agx_u32 off = X - r.t, q = r.q + off;
        const bool in_run = X >= r.t && off < r.n, is_last = q == js, is_src = q < js;
        const bool hit_run = in_run && (is_last || is_src);
        const bool end_of_run = off + 1 >= r.n;
        const bool direct = has_nx && (nx.q == q + 1 || nx.t == X + 1);       // deletion AG:1822-1838 / insertion AG:1707-1727; else a chain starts (AG:1736)
        // X between this run and the next: a read insertion next to a reference gap walks through these positions (AG:1737-1749)
        const agx_u32 p = r.q + r.n - 1;
        const bool in_gap = has_nx && X >= r.t + r.n && X < nx.t && nx.q >= p + 2 && p < lim;
        a.has = (hit_run || in_gap) ? 1u : a.has;
        a.type = hit_run ? (is_last ? (agx_u32)AGX_AT_K2ONLY : (agx_u32)AGX_AT_K1) : (in_gap ? (agx_u32)AGX_AT_CHAIN : a.type);
        a.q = hit_run ? q : a.q;
        a.slen = hit_run ? (is_last ? ((L - q) < k ? (L - q) : k) : k) : a.slen;
        a.has_succ = hit_run ? (is_last ? 0u : 1u) : (in_gap ? 1u : a.has_succ);
        a.xs = (hit_run && end_of_run && direct) ? nx.t : a.xs;
        qsucc = hit_run ? (is_last ? AGX_NONE : (end_of_run ? (direct ? nx.q : AGX_NONE) : q + 1)) : (in_gap ? ((X + 1 == nx.t) ? nx.q : AGX_NONE) : qsucc);
        r = nx;
    }
    const agx_u32 p0 = agx_pos_b(d, runs, a.q), p0s = agx_pos_b(d, runs, qsucc);
    a.p0 = (a.has && a.type != AGX_AT_CHAIN) ? p0 : AGX_NONE;
    a.p0s = qsucc != AGX_NONE ? p0s : AGX_NONE;
    return a;
}

// ---- hit_prep: orientation, left-mate choice, multi-hit suppression, arrival span (AG:1648-1679) -------------

AGX_HD agx_u32 agx_idx0_pos(const agx_hit &h, const agx_run *runs) {     // positionSets[hit][0] of mate1
    if (h.nruns1 == 0) return h.pos1;
    const agx_run r = runs[h.runs1];
    return r.q == 0 ? r.t : AGX_NONE;
}

// A later hit of a pair whose mate1 lands within a read length of an earlier hit of the same pair is dropped (AG:1650-1655).  hits[] is in
// file order (the hits of a pair are neighbours): hit_prep evaluates it per hit.
template <class GET> AGX_HD bool agx_hit_dup_by(GET hit_at, const agx_run *runs, agx_u32 h) {      // hit_at(i): hit i (the device unpacks the wire format on the way)
    const agx_hit H = hit_at(h);
    const agx_u32 me = agx_idx0_pos(H, runs);
    for (agx_u32 e = 1; e <= H.back; e++)
        if (agx_absdiff(me, agx_idx0_pos(hit_at(h - e), runs)) < (int)H.len) return true;
    return false;
}
AGX_HD bool agx_hit_dup(const agx_hit *hits, const agx_run *runs, agx_u32 h) { return agx_hit_dup_by([hits](agx_u32 i) { return hits[i]; }, runs, h); }

// Which mate is the LEFT one ("a": the mate whose aligned indices emit the events, AG:1644, 1672-1679)?  true = mate2: some read index below
// L - k is aligned in both mates with mate1's position beyond mate2's.  Decided where the arrays are packed (the engine's staging): only
// the a mate's bases are ever read by the build, so only they cross PCIe, in one row per (pair, a mate) — the staged hit carries the
// row in slot1 and this verdict in pad[0].
AGX_HD bool agx_hit_left_is_mate2(const agx_hit &H, const agx_run *runs, agx_u32 k) {
    const agx_u32 L = H.len;
    if (L <= k) return false;
    const agx_u32 lim = L - k;
    const agx_u32 n1 = H.nruns1 ? H.nruns1 : 1u, n2 = H.nruns2 ? H.nruns2 : 1u;
    for (agx_u32 i = 0; i < n1; i++) {
        const agx_run r1 = H.nruns1 ? runs[H.runs1 + i] : agx_run{0u, H.pos1, L};
        for (agx_u32 j = 0; j < n2; j++) {
            const agx_run r2 = H.nruns2 ? runs[H.runs2 + j] : agx_run{0u, H.pos2, L};
            agx_u32 lo = r1.q > r2.q ? r1.q : r2.q, hi1 = r1.q + r1.n, hi2 = r2.q + r2.n;
            agx_u32 hi = hi1 < hi2 ? hi1 : hi2; if (hi > lim) hi = lim;
            if (lo < hi && (r1.t + (lo - r1.q)) > (r2.t + (lo - r2.q))) return true;   // difference is constant on the overlap
        }
    }
    return false;
}

// returns 0 ok, 1 = same-strand mates ("BOWTIE ALIGNMENT ERROR", AG:1667-1671).  swap: agx_hit_left_is_mate2; a_slot: where the a mate's
// bases are (the device: the staged row; the test executor: the read slot)
AGX_HD int agx_hit_prep(const agx_hit &H, bool dup, bool swap, agx_u32 a_slot, const agx_run *runs, agx_u32 k, agx_dhit &d) {
    d.flags = 0; d.len = H.len; d.jstar = 0xFFFF; d.x_lo = 1; d.x_hi = 0;
    d.a_t0 = d.b_t0 = d.a_runs = d.b_runs = d.a_slot = 0; d.a_nruns = d.b_nruns = 0;
    if (dup) { d.flags = AGX_HF_SKIP; return 0; }
    if (H.rev1 == H.rev2) { d.flags = AGX_HF_SKIP; return 1; }
    const agx_u32 L = H.len;
    if (L <= k) { d.flags = AGX_HF_SKIP; return 0; }
    const agx_u32 lim = L - k;
    d.a_slot = a_slot;
    if (!swap) {
        d.a_t0 = H.pos1; d.b_t0 = H.pos2; d.a_runs = H.runs1; d.b_runs = H.runs2; d.a_nruns = H.nruns1; d.b_nruns = H.nruns2;
        if (H.rev1) d.flags |= AGX_HF_AREV;
    } else {
        d.a_t0 = H.pos2; d.b_t0 = H.pos1; d.a_runs = H.runs2; d.b_runs = H.runs1; d.a_nruns = H.nruns2; d.b_nruns = H.nruns1;
        if (H.rev2) d.flags |= AGX_HF_AREV;
    }
    // arrival span and the K2ONLY index.  Sources are the aligned indices below cut = min(lim, last aligned index);
    // jstar = first aligned index >= cut (a read whose last aligned index is below lim ends there, e.g. a trailing soft clip)
    const agx_u32 nr = d.a_nruns ? d.a_nruns : 1u;
    agx_u32 q_last = 0; bool any = false;
    for (agx_u32 i = 0; i < nr; i++) { const agx_run r = d.a_nruns ? runs[d.a_runs + i] : agx_run{0u, d.a_t0, L}; if (r.n) { q_last = r.q + r.n - 1; any = true; } }
    if (!any) { d.flags |= AGX_HF_SKIP; return 0; }
    const agx_u32 cut = lim < q_last ? lim : q_last;
    agx_u32 first_x = 0, last_x = 0; bool below = false, have_first = false;
    for (agx_u32 i = 0; i < nr; i++) {
        const agx_run r = d.a_nruns ? runs[d.a_runs + i] : agx_run{0u, d.a_t0, L};
        if (r.n == 0) continue;
        if (!have_first) { first_x = r.t; have_first = true; }
        if (r.q < cut) below = true;
        if (r.q + r.n > cut) {                              // this run holds the first aligned index >= cut
            const agx_u32 j = r.q >= cut ? r.q : cut;
            d.jstar = (agx_u16)j; last_x = r.t + (j - r.q);
            break;
        }
    }
    if (!below) { d.flags |= AGX_HF_SKIP; return 0; }       // no source index: the hit emits no event
    d.x_lo = first_x; d.x_hi = last_x;
    if (d.a_nruns == 0 && d.b_nruns == 0) { d.a_runs = (agx_u32)d.jstar << 16; d.b_runs = 0; }      // one linear piece (agx_decode_arrival): qs = 0, qe = jstar
    return 0;
}

// The record of hit d in the list of tile `tile`: d itself, or — where the hit's arrivals inside the tile are ONE linear piece — that piece
// in the form agx_decode_arrival decodes with a handful of scalar operations (three hits in eight have an indel or a soft clip in one of
// their mates, but a break point lies in one tile: the other tiles the hit touches see a single run of the a mate against a single run, or
// a hole, of the b mate).  A piece must give, for every position of the tile, exactly what the general decode gives: the same index, type
// and mate position, the successor on the next position with the next mate position.
// (agx_tile_piece hands the piece back as scalars: a record chosen between two structs is a choice between two addresses to the compiler, and both then live in scratch
// memory on the device — r05 measured agx_k_tile_fill at 0.72 ms that way against 0.51)
// (the record's fields come as values: a record that exists as a struct in a kernel's registers and is then chosen from — `runs ? runs[i].t : d.a_t0` — is a choice
// between two addresses to the compiler, and the struct then lives in scratch memory on the device)
AGX_HD bool agx_tile_piece_v(agx_u32 d_a_t0, agx_u32 d_b_t0, agx_u32 d_a_runs, agx_u32 d_b_runs, agx_u32 L, agx_u32 js, agx_u32 d_a_nruns, agx_u32 d_b_nruns, agx_u32 d_flags, agx_u32 d_x_lo, agx_u32 d_x_hi,
                             const agx_run *runs, agx_u32 tile, agx_u32 k, agx_u32 &p_a_t0, agx_u32 &p_b_t0, agx_u32 &p_a_runs, bool &p_bnone) {
    p_a_t0 = 0; p_b_t0 = 0; p_a_runs = 0; p_bnone = false;
    if ((d_a_nruns == 0 && d_b_nruns == 0) || (d_flags & AGX_HF_SKIP)) return false;
    const agx_u32 t0 = tile * AGX_TILE;
    const agx_u32 xs = d_x_lo > t0 ? d_x_lo : t0, xe = d_x_hi < t0 + AGX_TILE - 1u ? d_x_hi : t0 + AGX_TILE - 1u;
    if (xs > xe || js == 0xFFFFu || L <= k) return false;
    // the a run that holds xs must hold xe, and xe + 1 too if xe is an event source (its successor is then the next position)
    const agx_u32 na = d_a_nruns ? d_a_nruns : 1u;
    agx_u32 r_q = 0, r_t = 0, r_n = 0; bool found = false;
    for (agx_u32 i = 0; i < na; i++) {
        agx_u32 c_q = 0, c_t = d_a_t0, c_n = L;
        if (d_a_nruns) { const agx_run c = runs[d_a_runs + i]; c_q = c.q; c_t = c.t; c_n = c.n; }
        if (c_n && xs >= c_t && xs - c_t < c_n) { r_q = c_q; r_t = c_t; r_n = c_n; found = true; }
    }
    if (!found || xe - r_t >= r_n) return false;
    const agx_u32 qs = r_q + (xs - r_t), qe = r_q + (xe - r_t);
    if (qe > js) return false;                              // (cannot happen: x_hi is jstar's position)
    if (qe < js && xe + 1u - r_t >= r_n) return false;
    const agx_u32 qe2 = qe < js ? qe + 1u : qe;             // the last index whose mate position is looked at
    // the b mate over qs .. qe2: one run, or nothing at all
    const agx_u32 nb = d_b_nruns ? d_b_nruns : 1u;
    agx_u32 rb_q = 0, rb_t = 0, rb_n = 0; bool in_b = false, touches = false;
    for (agx_u32 i = 0; i < nb; i++) {
        agx_u32 c_q = 0, c_t = d_b_t0, c_n = L;
        if (d_b_nruns) { const agx_run c = runs[d_b_runs + i]; c_q = c.q; c_t = c.t; c_n = c.n; }
        if (!c_n) continue;
        if (qs >= c_q && qs - c_q < c_n) { rb_q = c_q; rb_t = c_t; rb_n = c_n; in_b = true; }
        if (c_q <= qe2 && c_q + c_n > qs) touches = true;
    }
    if (in_b && qe2 - rb_q >= rb_n) return false;
    if (!in_b && touches) return false;
    p_a_t0 = r_t - r_q; p_b_t0 = in_b ? rb_t - rb_q : 0u; p_a_runs = qs | (qe << 16); p_bnone = !in_b;
    return true;
}
AGX_HD bool agx_tile_piece(const agx_dhit &d, const agx_run *runs, agx_u32 tile, agx_u32 k, agx_u32 &p_a_t0, agx_u32 &p_b_t0, agx_u32 &p_a_runs, bool &p_bnone) {
    return agx_tile_piece_v(d.a_t0, d.b_t0, d.a_runs, d.b_runs, d.len, d.jstar, d.a_nruns, d.b_nruns, d.flags, d.x_lo, d.x_hi, runs, tile, k, p_a_t0, p_b_t0, p_a_runs, p_bnone);
}
AGX_HD agx_dhit agx_tile_record(const agx_dhit &d, const agx_run *runs, agx_u32 tile, agx_u32 k) {
    agx_u32 a_t0, b_t0, a_runs; bool bnone;
    agx_dhit p = d;
    if (agx_tile_piece(d, runs, tile, k, a_t0, b_t0, a_runs, bnone)) { p.a_t0 = a_t0; p.b_t0 = b_t0; p.a_runs = a_runs; p.b_runs = 0; p.a_nruns = p.b_nruns = 0; if (bnone) p.flags |= AGX_HF_BNONE; }
    return p;
}
// (r05's 32-byte tile records were agx_tile_record() as two 16-byte halves; r06's tile lists hold lean records — below.  agx_tile_record() stays: the test executor sweeps with it and
// checks that a piece decodes like the hit.)

// ---- lean tile records (r06): what pass 0 of the node sweep reads per list entry ----------------------------------------------------------------
// 32 bytes that describe a hit's arrivals INSIDE ONE TILE as one or two runs of lanes, each with its own pair of offsets: on lane l of a piece the arrival's read index
// is l + qoff and its mate position l + boff (or none).  One piece = what agx_tile_record() calls a linear piece (five entries in six).  TWO pieces cover most of the rest —
// the tile that holds a hit's one break point: an insertion or a deletion in the left mate (the pieces differ in qoff; behind a deletion the lanes between them have no
// arrival and the last lane of piece 1 steps over them: JUMP1), a break in the other mate (the pieces differ in boff; with MID the lanes between them have arrivals without a
// mate position — bases the other mate has inserted or clipped).  What fits neither (a third piece, a read insertion next to a reference gap with its CHAIN arrivals,
// empty runs) is kind GENERAL: the sweep then decodes the hit's derived record, dhit[hit], the way every other reader of a tile list does (the wider passes, the edge
// build's pass B: they only look at `hit`).  agx_lean_make() must produce a record only if agx_lean_decode() gives, on every lane of the tile, exactly the arrival
// agx_decode_arrival() gives for the hit — tests/hostsim checks that for every entry of every list it makes.
enum { AGX_LK_GENERAL = 0, AGX_LK_ONE = 1, AGX_LK_ONEX = 2, AGX_LK_TWO = 3 };      // ONE: one piece WITH mate positions and without a jump at its end — nothing to look at but lo1, span1, qoff1, boff1; ONEX: one piece with either; TWO: two pieces
enum { AGX_LF_AREV = 1u << 24, AGX_LF_BN1 = 1u << 25, AGX_LF_BN2 = 1u << 26, AGX_LF_JUMP1 = 1u << 27, AGX_LF_MID = 1u << 28, AGX_LF_JUMP2 = 1u << 29 };
struct agx_lrec {
    agx_u32 qoff1, boff1, qoff2, boff2;      // (boff unused where BN1 / BN2 says the piece has no mate positions)
    agx_u32 slot, lenjs;                     // read slot of the left mate; read length | jstar << 16
                                             // kind ONE only (it has no second piece): qoff2 = what the stored index of a lane's base is counted from — (lane ^ -reverse) + qoff2 —, boff2 = the lane of the K2ONLY arrival, jstar - qoff1 (any value: no such lane)
    agx_u32 geo;                             // lo1 | span1 << 6 | lo2 << 12 | span2 << 18 (a piece = lanes lo .. lo + span) | AGX_LF_* | kind << 30
    agx_u32 hit;                             // the hit's place in the tile order: dhit[hit]
};
struct agx_larr { agx_u32 has, last, q, p0, jump; };      // last: the K2ONLY arrival; jump: a K1 arrival whose successor is not position + 1 (every other K1 arrival's is)
AGX_HD agx_larr agx_lean_decode(const agx_lrec &r, agx_u32 lane) {
    const agx_u32 g = r.geo, kind = g >> 30;
    const agx_u32 lo1 = g & 63u, sp1 = (g >> 6) & 63u, lo2 = (g >> 12) & 63u, sp2 = (g >> 18) & 63u, js = r.lenjs >> 16;
    const bool in1 = lane - lo1 <= sp1, in2 = kind == AGX_LK_TWO && lane - lo2 <= sp2, mid = (g & AGX_LF_MID) && lane > lo1 + sp1 && lane < lo2;
    agx_larr a;
    a.has = (kind != AGX_LK_GENERAL && (in1 || in2 || mid)) ? 1u : 0u;
    a.q = lane + (in2 ? r.qoff2 : r.qoff1);
    a.p0 = in2 ? ((g & AGX_LF_BN2) ? AGX_NONE : lane + r.boff2) : (mid || (g & AGX_LF_BN1)) ? AGX_NONE : lane + r.boff1;
    a.last = a.q == js ? 1u : 0u;
    a.jump = (!a.last && (((g & AGX_LF_JUMP1) && lane == lo1 + sp1) || ((g & AGX_LF_JUMP2) && in2 && lane == lo2 + sp2))) ? 1u : 0u;
    return a;
}
// the mate position of read indices qa .. qb as sections of constant offset: how many (0 = more than three, or runs that overlap), and for section i its first index,
// whether it has no mate positions and, if it has, t - q of its run.  (No arrays: a table indexed by a running count lives in scratch memory on the device.  Runs come as
// values: a run chosen between a table entry and the simple mate's pseudo-run must not be a choice between addresses — agx_tile_piece_v.)
struct agx_bsec { agx_u32 n, q1, q2, none0, none1, none2, off0, off1, off2; };
#define AGX_LEAN_MAXRUNS 3u      // a mate with more runs than this is kind GENERAL: the runs are fetched up front, side by side (a loop that loads run after run is a chain of waits: agx_k_tile_fill's time)
AGX_HD agx_bsec agx_lean_bsections(agx_u32 b0q, agx_u32 b0t, agx_u32 b0n, agx_u32 b1q, agx_u32 b1t, agx_u32 b1n, agx_u32 b2q, agx_u32 b2t, agx_u32 b2n, agx_u32 nb, agx_u32 qa, agx_u32 qb) {      // (runs as values, never as structs that are chosen between)
    // (section i's fields are separate variables, each updated by a select on the running count: stores chosen by the count into a struct's fields become ONE store through a
    // computed address, and the struct then lives in scratch memory on the device)
    agx_u32 n = 0, q1 = 0, q2 = 0, none0 = 1u, none1 = 1u, none2 = 1u, off0 = 0, off1 = 0, off2 = 0;
    auto push = [&](bool on, agx_u32 q0, agx_u32 none, agx_u32 off) {
        const bool a0 = on && n == 0u, a1 = on && n == 1u, a2 = on && n == 2u;
        none0 = a0 ? none : none0; off0 = a0 ? off : off0;
        q1 = a1 ? q0 : q1; none1 = a1 ? none : none1; off1 = a1 ? off : off1;
        q2 = a2 ? q0 : q2; none2 = a2 ? none : none2; off2 = a2 ? off : off2;
        n += on ? 1u : 0u;
    };
    agx_u32 next = qa;                        // first index not yet assigned to a section
    bool bad = false;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (agx_u32 i = 0; i < AGX_LEAN_MAXRUNS; i++) {
        const agx_u32 cq = i == 0u ? b0q : i == 1u ? b1q : b2q, ct = i == 0u ? b0t : i == 1u ? b1t : b2t, cn = i == 0u ? b0n : i == 1u ? b1n : b2n;
        const bool touches = i < nb && cn != 0u && cq + cn > qa && cq <= qb && next <= qb;      // a run that holds some of what is left of qa .. qb
        const agx_u32 from = cq > qa ? cq : qa;
        bad = bad || (touches && from < next);                              // runs that overlap, or out of order
        push(touches && from > next, next, 1u, 0u);
        push(touches, from, 0u, ct - cq);
        next = touches ? cq + cn : next;                                    // (may lie beyond qb)
    }
    push(next <= qb, next, 1u, 0u);
    agx_bsec s; s.n = (bad || n > 3u) ? 0u : n; s.q1 = q1; s.q2 = q2; s.none0 = none0; s.none1 = none1; s.none2 = none2; s.off0 = off0; s.off1 = off1; s.off2 = off2;
    return s;
}
// the record of the hit with derived record (d_*) in the list of `tile`; hit = its place in the tile order
AGX_HD agx_lrec agx_lean_make_v(agx_u32 d_a_t0, agx_u32 d_b_t0, agx_u32 d_a_runs, agx_u32 d_b_runs, agx_u32 d_a_slot, agx_u32 L, agx_u32 js, agx_u32 d_a_nruns, agx_u32 d_b_nruns,
                                agx_u32 d_flags, agx_u32 d_x_lo, agx_u32 d_x_hi, const agx_run *runs, agx_u32 tile, agx_u32 k, agx_u32 hit) {
    agx_lrec r; r.qoff1 = r.boff1 = r.qoff2 = r.boff2 = 0; r.slot = d_a_slot; r.lenjs = L | (js << 16); r.hit = hit;
    r.geo = (d_flags & AGX_HF_AREV) ? (agx_u32)AGX_LF_AREV : 0u;      // kind GENERAL (the read slot, the lengths and the strand are there whatever the kind)
    const agx_u32 T0 = tile * AGX_TILE;
    const agx_u32 xs = d_x_lo > T0 ? d_x_lo : T0, xe = d_x_hi < T0 + AGX_TILE - 1u ? d_x_hi : T0 + AGX_TILE - 1u;
    if ((d_flags & AGX_HF_SKIP) || xs > xe || js == 0xFFFFu || L <= k || d_a_nruns > AGX_LEAN_MAXRUNS || d_b_nruns > AGX_LEAN_MAXRUNS) return r;
    // both mates' runs, all at once (a simple mate is one run of the whole read at its offset)
    const agx_u32 na = d_a_nruns ? d_a_nruns : 1u, nb = d_b_nruns ? d_b_nruns : 1u;
    agx_u32 a0q = 0, a0t = d_a_t0, a0n = L, a1q = 0, a1t = 0, a1n = 0, a2q = 0, a2t = 0, a2n = 0, b0q = 0, b0t = d_b_t0, b0n = L, b1q = 0, b1t = 0, b1n = 0, b2q = 0, b2t = 0, b2n = 0;
    if (d_a_nruns) { const agx_run c = runs[d_a_runs]; a0q = c.q; a0t = c.t; a0n = c.n; }
    if (d_a_nruns > 1u) { const agx_run c = runs[d_a_runs + 1u]; a1q = c.q; a1t = c.t; a1n = c.n; }
    if (d_a_nruns > 2u) { const agx_run c = runs[d_a_runs + 2u]; a2q = c.q; a2t = c.t; a2n = c.n; }
    if (d_b_nruns) { const agx_run c = runs[d_b_runs]; b0q = c.q; b0t = c.t; b0n = c.n; }
    if (d_b_nruns > 1u) { const agx_run c = runs[d_b_runs + 1u]; b1q = c.q; b1t = c.t; b1n = c.n; }
    if (d_b_nruns > 2u) { const agx_run c = runs[d_b_runs + 2u]; b2q = c.q; b2t = c.t; b2n = c.n; }
    // the left mate's runs that hold arrivals of this tile: at most two, no empty run anywhere (an empty run is still the "next run" of the one before it)
    agx_u32 np = 0, lo0 = 0, hi0 = 0, qoff0 = 0, jump0 = 0, lo1 = 0, hi1 = 0, qoff1 = 0, jump1 = 0;
    bool bad = false;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (agx_u32 i = 0; i < AGX_LEAN_MAXRUNS; i++) {
        const agx_u32 cq = i == 0u ? a0q : i == 1u ? a1q : a2q, ct = i == 0u ? a0t : i == 1u ? a1t : a2t, cn = i == 0u ? a0n : i == 1u ? a1n : a2n;
        const agx_u32 nxq = i == 0u ? a1q : a2q, nxt = i == 0u ? a1t : a2t;      // (the next run: only looked at where has_nx)
        const bool on = i < na;
        const agx_u32 c_end = ct + cn - 1u;
        const bool has_nx = i + 1u < na;
        const bool direct_q = has_nx && nxq == cq + cn, direct_t = has_nx && nxt == c_end + 1u;
        bad = bad || (on && cn == 0u);
        // CHAIN arrivals between this run and the next (agx_decode_arrival's in_gap) that fall into the tile: not a lean record
        bad = bad || (on && has_nx && !direct_q && !direct_t && nxt > c_end + 1u && c_end + 1u <= T0 + AGX_TILE - 1u && nxt - 1u >= T0);
        const bool holds = on && cn != 0u && !(c_end < xs || ct > xe);
        bad = bad || (holds && np >= 2u);
        const agx_u32 lo = ct > xs ? ct : xs, hi = c_end < xe ? c_end : xe;
        // the piece ends where the run ends, on an event source (an index below jstar): where does its successor go?  To the next run's first index — on position + 1 (a read
        // insertion, or a run cut in two: nothing special), over a gap of the reference (a read deletion: a JUMP), or, if the next run continues neither in the read nor on
        // the reference, through CHAIN arrivals: not a lean record.  (No next run cannot happen: the read's last aligned index is not below jstar.)
        const bool ends = holds && hi == c_end && cq + cn - 1u < js;
        bad = bad || (ends && (!has_nx || (!direct_q && !direct_t)));
        const agx_u32 jump = (ends && !direct_t) ? 1u : 0u;
        const bool p0 = holds && np == 0u, p1 = holds && np == 1u;
        lo0 = p0 ? lo - T0 : lo0; hi0 = p0 ? hi - T0 : hi0; qoff0 = p0 ? cq - ct + T0 : qoff0; jump0 = p0 ? jump : jump0;
        lo1 = p1 ? lo - T0 : lo1; hi1 = p1 ? hi - T0 : hi1; qoff1 = p1 ? cq - ct + T0 : qoff1; jump1 = p1 ? jump : jump1;
        np += holds ? 1u : 0u;
    }
    if (bad || np == 0u || (np == 2u && lo1 <= hi0)) return r;
    agx_u32 geo = r.geo;
    // (the sections under the first piece once, for both shapes: the function is inlined where it is called)
    const agx_bsec s = agx_lean_bsections(b0q, b0t, b0n, b1q, b1t, b1n, b2q, b2t, b2n, nb, lo0 + qoff0, hi0 + qoff0);
    if (np == 2u) {
        // two pieces of the left mate: the other mate must be one section under each
        const agx_bsec s2 = agx_lean_bsections(b0q, b0t, b0n, b1q, b1t, b1n, b2q, b2t, b2n, nb, lo1 + qoff1, hi1 + qoff1);
        if (s.n != 1u || s2.n != 1u) return r;
        r.qoff1 = qoff0; r.qoff2 = qoff1; r.boff1 = qoff0 + s.off0; r.boff2 = qoff1 + s2.off0;
        geo |= (s.none0 ? (agx_u32)AGX_LF_BN1 : 0u) | (s2.none0 ? (agx_u32)AGX_LF_BN2 : 0u) | (jump0 ? (agx_u32)AGX_LF_JUMP1 : 0u) | (jump1 ? (agx_u32)AGX_LF_JUMP2 : 0u);
        r.geo = geo | lo0 | ((hi0 - lo0) << 6) | (lo1 << 12) | ((hi1 - lo1) << 18) | ((agx_u32)AGX_LK_TWO << 30);
        return r;
    }
    // one piece of the left mate: the other mate in one, two, or three sections of which the middle one has no positions
    if (s.n == 0u || (s.n == 3u && !(s.none1 && !s.none0 && !s.none2))) return r;
    r.qoff1 = r.qoff2 = qoff0; r.boff1 = qoff0 + s.off0;
    geo |= s.none0 ? (agx_u32)AGX_LF_BN1 : 0u;
    if (s.n == 1u) {
        const bool plain = !jump0 && !s.none0;
        if (plain) { r.qoff2 = (d_flags & AGX_HF_AREV) ? L - qoff0 : qoff0; r.boff2 = js - qoff0; }      // (what the sweep's arm for this kind reads instead of unpacking and subtracting: see agx_lrec)
        r.geo = geo | lo0 | ((hi0 - lo0) << 6) | (jump0 ? (agx_u32)AGX_LF_JUMP1 : 0u) | ((agx_u32)(plain ? AGX_LK_ONE : AGX_LK_ONEX) << 30); return r;
    }
    const agx_u32 q_last = s.n == 3u ? s.q2 : s.q1, none_last = s.n == 3u ? s.none2 : s.none1, off_last = s.n == 3u ? s.off2 : s.off1;
    const agx_u32 l2 = q_last - qoff0, h1 = s.q1 - qoff0 - 1u;              // first lane of the last section, last lane of the first
    r.boff2 = qoff0 + off_last;
    geo |= (none_last ? (agx_u32)AGX_LF_BN2 : 0u) | (s.n == 3u ? (agx_u32)AGX_LF_MID : 0u) | (jump0 ? (agx_u32)AGX_LF_JUMP2 : 0u);      // (the run's end is piece 2's end)
    r.geo = geo | lo0 | ((h1 - lo0) << 6) | (l2 << 12) | ((hi0 - l2) << 18) | ((agx_u32)AGX_LK_TWO << 30);
    return r;
}
AGX_HD agx_lrec agx_lean_make(const agx_dhit &d, const agx_run *runs, agx_u32 tile, agx_u32 k, agx_u32 hit) {
    return agx_lean_make_v(d.a_t0, d.b_t0, d.a_runs, d.b_runs, d.a_slot, d.len, d.jstar, d.a_nruns, d.b_nruns, d.flags, d.x_lo, d.x_hi, runs, tile, k, hit);
}

// conti-mer head of position x (upload-time kernel / test executor).  The table has n_pos + 1 entries: entry n_pos is the head of "no
// position" (no conti-mers), which the node sweep loads for an arrival without a mate position instead of selecting afterwards.
AGX_HD void agx_cm_head_pos(const agx_u32 *cm_start, const agx_cmkey *cm, agx_cmhead *head, agx_u32 x, agx_u32 n_pos) {
    if (x == n_pos) { head[x] = agx_cmhead{AGX_NONE, AGX_NONE, 0u, 0u}; return; }
    const agx_u32 s = cm_start[x], n = cm_start[x + 1] - s;
    head[x] = n ? agx_cmhead{cm[s].cid, cm[s].coff, n, s} : agx_cmhead{AGX_NONE, AGX_NONE, 0u, s};
}

// ---- conti-mer threads as runs ------------------------------------------------------------------------------------------------
// A contig placement threads one conti-mer per position along consecutive positions (AG:884-1217), with the contig offset moving in
// step, until an indel makes it jump.  So the conti-mer tables cross PCIe as RUNS (40 bytes per run instead of 4 bytes per position and 8
// per conti-mer): element j of a run is the rank-th conti-mer of position pos0 + j, (cid, coff0 + j * dcoff).  The runs also carry what
// the walk needs when it leaves the k-mer graph at one of their positions (agx_hop): following the chain from element j appends
// chain_str[hop_str0 + j, .. + hop_len0 - j) and lands on hop_end.  build_chains() (host) makes them; the device expands them at upload.
struct agx_cmseg { agx_u32 pos0, len, cid, coff0, dcoff, rank, hop_str0, hop_len0, hop_end, elem0; };   // elem0: elements of all earlier runs

// run that holds element e (runs in elem0 order)
AGX_HD agx_u32 agx_seg_of_elem(const agx_cmseg *segs, agx_u32 n_segs, agx_u32 e) {
    agx_u32 lo = 0, hi = n_segs;                          // last run with elem0 <= e
    while (hi - lo > 1) { const agx_u32 mid = lo + (hi - lo) / 2; if (segs[mid].elem0 <= e) lo = mid; else hi = mid; }
    return lo;
}
// The tables straight from the runs, without a count pass and a scan (r03): the host knows how many conti-mers every position carries (it threaded
// them), so it sends the COUNTS as runs — [pos0, pos0 + len) carry cnt conti-mers each, `base` before them — and both runs lists cut into chunks
// (run, offset) that a block takes without searching.  A position's cm_start and, where it is empty, its head come from its count run; the keys, and the
// head of every position that has conti-mers (written by its rank-0 element), from the conti-mer runs.
struct agx_cntrun { agx_u32 pos0, len, cnt, base; };
struct agx_chunk { agx_u32 run, off; };
#define AGX_CM_CHUNK 2048u
#define AGX_SEG_INDEX 1024u      // positions per entry of the run index (agx_compact_args::seg_index)
AGX_HD void agx_cm_layout_pos(const agx_cntrun &r, agx_u32 j, agx_u32 *cm_start, agx_cmhead *head) {      // position r.pos0 + j
    const agx_u32 x = r.pos0 + j, s = r.base + j * r.cnt;
    cm_start[x] = s;
    if (r.cnt == 0) head[x] = agx_cmhead{AGX_NONE, AGX_NONE, 0u, s};
}
AGX_HD void agx_cm_fill_elem(const agx_cmseg &g, agx_u32 j, const agx_u32 *cm_start, agx_cmkey *cm, agx_cmhead *head) {      // element j of run g
    const agx_u32 x = g.pos0 + j, s = cm_start[x], n = cm_start[x + 1] - s, coff = g.coff0 + j * g.dcoff;
    cm[s + g.rank] = agx_cmkey{g.cid, coff};
    if (g.rank == 0) head[x] = agx_cmhead{g.cid, coff, n, s};
}
struct agx_hop;
// hop entry of position x from the runs: defined where x carries exactly one conti-mer (its run has rank 0; the first n_seg0 runs are the
// rank-0 ones, sorted by pos0) that has a next
template <class HOP> AGX_HD HOP agx_seg_hop(const agx_cmseg *segs, agx_u32 n_seg0, const agx_u32 *cm_start, agx_u32 x) {
    HOP h; h.str_off = 0; h.len = 0; h.end_pos = 0;
    if (n_seg0 == 0 || cm_start[x + 1] - cm_start[x] != 1) return h;
    agx_u32 lo = 0, hi = n_seg0;                          // last rank-0 run with pos0 <= x
    while (hi - lo > 1) { const agx_u32 mid = lo + (hi - lo) / 2; if (segs[mid].pos0 <= x) lo = mid; else hi = mid; }
    const agx_cmseg g = segs[lo];
    const agx_u32 j = x - g.pos0;
    if (x < g.pos0 || j >= g.len || j >= g.hop_len0) return h;      // (j == hop_len0: the chain's last conti-mer, nothing to follow)
    h.str_off = g.hop_str0 + j; h.len = g.hop_len0 - j; h.end_pos = g.hop_end;
    return h;
}

// ---- node sweep: one lane = one position --------------------------------------------------------------------

struct agx_sweep_args {
    // static graph inputs
    const agx_u32 *cm_start;      // [n_pos+1] conti-mers of position x are cm[cm_start[x] .. cm_start[x+1])
    const agx_cmkey *cm;
    const agx_cmhead *cm_head;    // [n_pos + 1], the last one empty (agx_cm_head_pos)
    const char *ref;              // [n_pos] reference base per position (incl. appended positions)
    // reads
    const agx_dhit *dhit;
    const agx_run *runs;
    const agx_u8 *vcodes; agx_u32 stride; // vote codes (agx_vote_code) of the read bases: read slot s starts at vcodes + s*stride
    // tile lists
    const agx_u32 *tile_off;      // [n_tiles+1]
#if defined(__HIPCC__)
    const uint4 *tile_recs;       // [entries][2] first 32 bytes of the derived record of every list entry, hits in SAM order inside a tile
                                  // (device only: the kernels' scalar record stream; the test executor reads dhit through its own lists)
#else
    const void *tile_recs;
#endif
    agx_u32 n_pos, n_tiles, k; int iv; int coverage;
    // node table
    agx_u32 *node_start;          // [n_pos]
    agx_u16 *node_cnt;            // [n_pos] (16 bits: the reference's vector<KMer> is unbounded, AG:1375-1390; up to AGX_MAXV_HUGE variants are held, and 63 positions' side ids must fit side_pk's low half)
    agx_u8 *pos_succ;             // [n_pos] bit 0: some arrival at x steps to x+1 (written by the node sweep, read by the edge build)
    agx_u32 *side_pk;             // [n_pos] written with the nodes: surviving variants beyond the first ("side ids") of this position << 16 | side ids of the
                                  // tile's earlier positions; tile_side[tile] = side ids of the whole tile (input of the walk preparation's scan)
    agx_u32 *tile_side;
    agx_u32 *nk_cid, *nk_coff, *nk_cid0, *nk_coff0, *nk_off0;   // [pool]
    agx_u8 *n_base, *n_flags;     // consensus base ('X' = none: use the reference base, AG:1997-2001), AGX_NF_*
    agx_sref *n_sref;
    agx_u32 *n_next;              // [pool*AGX_MAXE]
    int *n_counts;                // optional [pool*6] cov,A,C,G,T,N (parity/debug), may be null
    agx_u32 pool_cap;
    agx_u32 *sweep_stats;         // profiling builds only (-DAGX_SWEEP_STATS): event counters of the node sweep, else null
};

// profiling hook of the node sweep: counts wavefronts in which some lane meets `cond` (slot i) and the lanes that do (slot i + 1)
#if defined(AGX_SWEEP_STATS) && defined(__HIP_DEVICE_COMPILE__)
#define AGX_STAT(A, i, cond) do { const unsigned long long m_ = __ballot(cond); if (m_ && (A).sweep_stats && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m_)) { atomicAdd((A).sweep_stats + (i), 1u); atomicAdd((A).sweep_stats + (i) + 1, (agx_u32)__popcll(m_)); } } while (0)
#else
#define AGX_STAT(A, i, cond) do { } while (0)
#endif

// What a read base votes for (updateKBases, AG:1340-1351, after reverseComplement, AG:854-865: only ACGT are complemented).
// A stored character (file orientation) has one of five CLASSES: A, C, G, T = 0..3, anything else = 4; (c>>1)&3 maps A,C,T,G to
// 0,1,2,3, so swapping 2 and 3 gives the class.  The packed-array boundary carries the classes, two per byte (4 bits each: the k-mer
// strings, which must keep every byte, stay on the host); the device expands them once per upload into VOTE CODES — low nibble = the
// bucket field a forward-strand read votes for with this base, high nibble = the field a reverse-strand read votes for — so that the
// sweep's vote is one bit-field extract.  The four base counters are consecutive fields in A, C, G, T order: complementing is 3 - class.
AGX_HD agx_u32 agx_base_class(agx_u32 c) {
    const agx_u32 idx = (c >> 1) & 3u;
    const bool acgt = c == ((0x47544341u >> (idx * 8u)) & 0xFFu);                 // "ACTG"[idx]
    return acgt ? (idx ^ (idx >> 1)) : 4u;
}
AGX_HD agx_u8 agx_class_vote_code(agx_u32 cls) {
    const agx_u32 fwd = cls < 4u ? (agx_u32)AGX_F_A + cls : (agx_u32)AGX_F_N, rev = cls < 4u ? (agx_u32)AGX_F_A + 3u - cls : (agx_u32)AGX_F_N;
    return (agx_u8)(fwd | (rev << 4));
}
AGX_HD agx_u8 agx_vote_code(agx_u32 c) { return agx_class_vote_code(agx_base_class(c)); }
// packed classes: base 2j of a read slot in the low nibble of byte j, base 2j+1 in the high nibble
AGX_HD agx_u8 agx_pack_classes(agx_u32 c_even, agx_u32 c_odd) { return (agx_u8)(agx_base_class(c_even) | (agx_base_class(c_odd) << 4)); }
// What crosses PCIe: TWO bits per base (the class of A, C, G, T; 0 for anything else), base j of a row in bits 2*(j & 3) of byte j >> 2,
// plus the list of the bases inside a read that are "anything else" (index into the vote-code array = row * stride + j): the upload-time
// kernel expands the 2-bit classes and then patches the listed bytes.  Reads carry few such bases; every one costs 8 bytes instead of half a bit.
AGX_HD agx_u8 agx_pack_classes2(agx_u32 c0, agx_u32 c1, agx_u32 c2, agx_u32 c3) {
    return (agx_u8)((agx_base_class(c0) & 3u) | ((agx_base_class(c1) & 3u) << 2) | ((agx_base_class(c2) & 3u) << 4) | ((agx_base_class(c3) & 3u) << 6));
}

// ---- read rows relative to the reference (upload form, r04) ----------------------------------------------------------------------------
// The 2-bit rows are more than half of what a unit sends over PCIe, and nearly all of their bases equal the reference base they are aligned
// to.  So a row crosses as the list of the bases that DIFFER from what the reference predicts for it, given the alignment of the row's
// ANCHOR: the first hit (file order) that names the row.  Its left mate's runs (q, t, n) put read indices q .. q + n - 1 on positions
// t .. t + n - 1 (a mate that is one full-length run: q = 0, t = the hit's offset, n = len); stored base j (file orientation) is read index j
// of a forward mate and len - 1 - j of a reverse mate, complemented; a base no run covers (inserted, clipped) is predicted as class 0.
// Per row one count byte (AGX_ROW_EXPLICIT: the row crosses as its 2-bit classes, as before — too many differences, an anchor whose runs
// do not fit the read or the unit) and a stream of 16-bit units: a difference = index << 2 | class; an explicit row = its stride / 4 bytes
// (rounded up to whole units).  Bases at and beyond the read's length are class 0 in both forms, so the expanded vote codes are the same
// bytes either way (tests/test_row_diffs.py).  The codec is exact for ANY anchor geometry: the host makes the differences against the
// very prediction the device will compute (agx_row_expected16, one function).
#define AGX_ROW_EXPLICIT 255u
#define AGX_ROW_MAXSTRIDE 256u                            // rows of longer reads keep the 2-bit form (agx_k_expand_rows holds 64 rows of vote codes in LDS; a difference's index has 14 bits)
#define AGX_ROW_MAXRUNS 32u                               // anchors with more runs than this keep their rows as they are
AGX_HD agx_u32 agx_row_explicit_units(agx_u32 stride) { return (stride / 4u + 1u) / 2u; }
AGX_HD agx_u32 agx_row_units(agx_u32 cnt, agx_u32 stride) { return cnt == AGX_ROW_EXPLICIT ? agx_row_explicit_units(stride) : cnt; }
// the geometry of a hit's left mate as the wire record has it: where its read index 0 sits if it is one full-length run, its strand, its runs if not
// (fields are read into values before anything is chosen between them: a choice between two members is a choice between two addresses to the compiler, and
// the record then lives in scratch memory on the device)
AGX_HD agx_u32 agx_whit_left_t0(const agx_whit &w) { const agx_u32 a = w.a, b = w.b, f = w.flags; return (f & AGX_WF_LEFT2) ? b : a; }
AGX_HD bool agx_whit_left_rev(const agx_whit &w) { const agx_u32 f = w.flags; return (f & ((f & AGX_WF_LEFT2) ? (agx_u32)AGX_WF_REV2 : (agx_u32)AGX_WF_REV1)) != 0; }
AGX_HD bool agx_whit_left_simple(const agx_whit &w) { const agx_u32 f = w.flags; return (f & ((f & AGX_WF_LEFT2) ? (agx_u32)AGX_WF_RUNS2 : (agx_u32)AGX_WF_RUNS1)) == 0; }
AGX_HD agx_u32 agx_whit_side(const agx_whit &w) { const agx_u32 a = w.a, b = w.b, f = w.flags; return (f & AGX_WF_RUNS1) ? a : b; }      // (of a hit with a multi-run mate)
AGX_HD agx_u32 agx_wside_left_first(const agx_whit &w, const agx_wside &sd) { const agx_u32 r1 = sd.runs1, r2 = sd.runs2, f = w.flags; return (f & AGX_WF_LEFT2) ? r2 : r1; }
AGX_HD agx_u32 agx_wside_left_count(const agx_whit &w, const agx_wside &sd) { const agx_u32 n = sd.nruns, f = w.flags; return (f & AGX_WF_LEFT2) ? n >> 16 : n & 0xFFFFu; }
// ---- hits in tile order (r05) ----------------------------------------------------------------------------------------------------------
// The build walks a unit's hits in the order of the tile their first arrival falls in (the staging makes the permutation: stage_order in agx_engine.cpp): neighbouring
// lanes of agx_k_hit_prep then want the same tiles' counters, and a tile's list is a filter over a WINDOW of that order (agx_k_tile_fill) instead of a scatter of
// 4-byte slot words all over HBM (r04: 893 MB written per 30 Mb unit where 250 were needed).  The key: the position of the left mate's first aligned base —
// what agx_hit_prep calls x_lo for every hit it keeps (a hit it drops has no arrivals: where it stands in the order does not matter).
AGX_HD agx_u32 agx_whit_first_x(const agx_whit &w, const agx_wside *sides, const agx_wrun *wruns) {
    if (agx_whit_left_simple(w)) return agx_whit_left_t0(w);
    const agx_wside sd = sides[agx_whit_side(w)];
    const agx_u32 f = agx_wside_left_first(w, sd), n = agx_wside_left_count(w, sd);
    for (agx_u32 i = 0; i < n; i++) if (wruns[f + i].n) return wruns[f + i].t;
    return 0u;
}
// tiles a tile's list looks back over: a hit's arrivals span len - k + 1 positions, i.e. at most 1 + ceil((len - k) / 64) tiles; hits that span more (long deletions)
// are few and go through a list of their own (AGX_LONG_MAX of them at most: beyond that the unit's lists are made the dense way, by scatter)
#define AGX_LOOKBACK_MAX 16u
#define AGX_LONG_MAX 1024u
AGX_HD agx_u32 agx_tile_lookback(agx_u32 maxlen, agx_u32 k) {
    const agx_u32 span = maxlen > k ? maxlen - k : 0u, lb = 1u + (span + AGX_TILE - 1u) / AGX_TILE;
    return lb < 2u ? 2u : lb > AGX_LOOKBACK_MAX ? AGX_LOOKBACK_MAX : lb;
}

// the 2-bit codes of positions p .. p + 15 of the packed reference (position p in the low bits); positions below 0 read as 0.  Reads the
// 32-bit words p >> 4 and (p >> 4) + 1: the buffer carries that much slack behind its last position.
AGX_HD agx_u32 agx_ref_window16(const agx_u32 *wref, long long p) {
    if (p <= -16) return 0u;
    if (p < 0) return wref[0] << (2u * (agx_u32)(-p));
    const size_t w = (size_t)(p >> 4); const agx_u32 sh = 2u * (agx_u32)(p & 15);
    const agx_u32 lo = wref[w], hi = wref[w + 1];
    return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
}
AGX_HD agx_u32 agx_reverse_pairs16(agx_u32 x) {          // the sixteen 2-bit groups of x in reverse order
    x = (x >> 16) | (x << 16);
    x = ((x & 0xFF00FF00u) >> 8) | ((x & 0x00FF00FFu) << 8);
    x = ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4);
    return ((x & 0xCCCCCCCCu) >> 2) | ((x & 0x33333333u) << 2);
}
// what the reference predicts for stored bases j0 .. j0 + 15 of a row (j0 a multiple of 16): 2-bit classes, base j0 in the low bits.  Every run must lie
// inside the read (q + n <= len) and inside the unit (t + n <= positions): the host checks that before it chooses this form for a row.
AGX_HD agx_u32 agx_row_run16(const agx_u32 *wref, const agx_wrun r, agx_u32 len, bool rev, agx_u32 j0) {      // one run's share of the chunk
    const agx_u32 lo = rev ? len - r.q - r.n : r.q, hi = lo + r.n;                         // the run's stored indices
    const agx_u32 a = lo > j0 ? lo : j0, b = hi < j0 + 16u ? hi : j0 + 16u;
    if (a >= b) return 0u;
    const long long base = (long long)r.t - (long long)r.q;                               // position of read index 0, were the run that long
    const agx_u32 win = rev ? ~agx_reverse_pairs16(agx_ref_window16(wref, base + (long long)len - 16ll - (long long)j0)) : agx_ref_window16(wref, base + (long long)j0);
    const agx_u32 nb = b - a, m = (nb < 16u ? (1u << (2u * nb)) - 1u : 0xFFFFFFFFu) << (2u * (a - j0));
    return win & m;
}
// left_runs / nruns = the runs of the anchor's left mate (nruns = 0: one full-length run at the hit's offset)
AGX_HD agx_u32 agx_row_expected16(const agx_u32 *wref, const agx_whit &w, const agx_wrun *left_runs, agx_u32 nruns, agx_u32 j0) {
    const agx_u32 len = w.len; const bool rev = agx_whit_left_rev(w);
    if (j0 >= len) return 0u;
    if (nruns == 0) return agx_row_run16(wref, agx_wrun{agx_whit_left_t0(w), (agx_u16)0, w.len}, len, rev, j0);
    agx_u32 c = 0;
    for (agx_u32 i = 0; i < nruns; i++) c |= agx_row_run16(wref, left_runs[i], len, rev, j0);
    return c;
}
// the classes of stored bases j0 .. j0 + 15 of a row out of its upload form: cnt = the row's count byte, units = its part of the stream, w = its anchor,
// left_runs / nruns = the runs of the anchor's left mate (nruns = 0: one full-length run at the hit's offset)
AGX_HD agx_u32 agx_row_chunk16(const agx_u32 *wref, const agx_u16 *units, agx_u32 cnt, const agx_whit &w, const agx_wrun *left_runs, agx_u32 nruns, agx_u32 j0, agx_u32 stride) {
    if (cnt == AGX_ROW_EXPLICIT) {
        const agx_u32 nu = agx_row_explicit_units(stride), i = j0 / 8u;      // (a unit holds eight bases)
        const agx_u32 lo = i < nu ? units[i] : 0u, hi = i + 1u < nu ? units[i + 1u] : 0u;
        return lo | (hi << 16);
    }
    agx_u32 c = agx_row_expected16(wref, w, left_runs, nruns, j0);
    for (agx_u32 i = 0; i < cnt; i++) {
        const agx_u32 e = units[i], j = e >> 2, sh = 2u * (j & 15u);
        if ((j >> 4) == (j0 >> 4)) c = (c & ~(3u << sh)) | ((e & 3u) << sh);
    }
    return c;
}
// the vote codes of four bases (classes in the low byte of c), one per byte
AGX_HD agx_u32 agx_vote_codes4(agx_u32 c) {
    agx_u32 o = 0;
    for (int b = 0; b < 4; b++) o |= (agx_u32)agx_class_vote_code((c >> (2 * b)) & 3u) << (8 * b);
    return o;
}
// The anchor of row 64 b + l, given the first anchor of the block (h0 = the anchor of row 64 b): rows are numbered in the order of their anchors (the host checks
// it, and that a block's anchors lie within 224 hits of its first, before it chooses this form), so it is the l-th set bit of the anchor bits at or behind h0.
// Reads up to eight 32-bit words from h0 >> 5 on.  NONE if there is no such bit there.
AGX_HD agx_u32 agx_anchor_select(const agx_u32 *bits, agx_u32 h0, agx_u32 l) {
    const agx_u32 w0 = h0 >> 5;
    agx_u32 need = l;
    for (agx_u32 i = 0; i < 8u; i++) {
        agx_u32 x = bits[w0 + i];
        if (i == 0) x &= ~0u << (h0 & 31u);
        const agx_u32 pc = (agx_u32)__builtin_popcount(x);
        if (need < pc) { for (agx_u32 k = 0; k < need; k++) x &= x - 1u; return (w0 + i) * 32u + (agx_u32)__builtin_ctz(x); }
        need -= pc;
    }
    return AGX_NONE;
}
// A whole row out of its upload form into `out` (stride bytes of vote codes, 4-byte aligned; the bases that are not A, C, G, T are patched in afterwards from their list):
// the prediction sixteen bases at a time, then one byte per difference.  What agx_k_expand_rows runs per lane (out = the lane's row in LDS) and the test executor per row.
AGX_HD void agx_row_decode(const agx_u32 *wref, const agx_u16 *units, agx_u32 cnt, const agx_whit &w, const agx_wrun *left_runs, agx_u32 nruns, agx_u32 stride, agx_u8 *out) {
    agx_u32 *o32 = (agx_u32 *)out;
    const bool expl = cnt == AGX_ROW_EXPLICIT;
    for (agx_u32 j0 = 0; j0 < stride; j0 += 16u) {
        const agx_u32 cls = expl ? agx_row_chunk16(wref, units, cnt, w, left_runs, nruns, j0, stride) : agx_row_expected16(wref, w, left_runs, nruns, j0);
        for (agx_u32 q = 0; q < 4u && j0 + 4u * q < stride; q++) o32[j0 / 4u + q] = agx_vote_codes4((cls >> (8u * q)) & 0xFFu);
    }
    if (expl) return;
    for (agx_u32 i = 0; i < cnt; i++) { const agx_u32 e = units[i], j = e >> 2; if (j < stride) out[j] = agx_class_vote_code(e & 3u); }
}

// first compatible variant or append (AG:1375-1390 / 1493-1506).  Returns the index, or NONE when the bucket is full.
AGX_HD agx_u32 agx_match_or_insert(const agx_bucket &b, agx_u32 &cnt, const agx_key &key, int iv, bool is_k1, agx_u32 s0, agx_u32 s1) {
    agx_u32 v = 0;
    for (; v < cnt; v++) if (agx_compatible(key, b, v, iv)) break;
    if (v == cnt) {
        if (cnt == b.maxv) return AGX_NONE;
        agx_b(b, v, AGX_F_CID) = key.cid; agx_b(b, v, AGX_F_COFF) = key.coff; agx_b(b, v, AGX_F_CID0) = key.cid0; agx_b(b, v, AGX_F_COFF0) = key.coff0;
        agx_b(b, v, AGX_F_OFF0) = key.off0; agx_cnt_init(b, v, 0u);
        agx_b(b, v, AGX_F_S0) = s0; agx_b(b, v, AGX_F_S1) = s1;
        cnt++;
    }
    if (is_k1) agx_cnt_add<false>(b, v, AGX_F_COV, 1u);
    return v;
}

// Candidate keys of an arrival at X with mate position p0 (AG:1369-1477): {conti-mers at X | none} x {conti-mers at p0 | none}, X-major.
// Calls f(key) for each; f returns false to stop.
template <class F>
AGX_HD void agx_for_candidates(const agx_sweep_args &A, agx_u32 cx_s, agx_u32 cx_n, agx_u32 p0, F f) {
    agx_u32 c0_s = 0, c0_n = 0;
    if (p0 != AGX_NONE) { c0_s = A.cm_start[p0]; c0_n = A.cm_start[p0 + 1] - c0_s; }
    const agx_u32 nx = cx_n ? cx_n : 1u, n0 = c0_n ? c0_n : 1u;
    for (agx_u32 i = 0; i < nx; i++) {
        agx_key key; key.off0 = p0;
        if (cx_n) { const agx_cmkey c = A.cm[cx_s + i]; key.cid = c.cid; key.coff = c.coff; } else { key.cid = AGX_NONE; key.coff = AGX_NONE; }
        for (agx_u32 j = 0; j < n0; j++) {
            if (c0_n) { const agx_cmkey c = A.cm[c0_s + j]; key.cid0 = c.cid; key.coff0 = c.coff; } else { key.cid0 = AGX_NONE; key.coff0 = AGX_NONE; }
            if (!f(key)) return;
        }
    }
}

// Everything one arrival needs from memory.  The sweep fetches it ONE HIT AHEAD of its use, so the dependent global loads
// (mate conti-mer range -> first entry, vote base) of hit i+1 are in flight while hit i updates the bucket in LDS.
struct agx_pre {
    agx_u32 has, type, p0, step1, jump;    // arrival present at this position; AGX_AT_*; mate position; its successor is position+1 / some other position
    agx_cmhead h; agx_u32 mate;            // conti-mer head of the mate position as loaded (position 0 if there is no mate), mate present
    agx_u32 s0, s1, cbyte, rev;            // k-mer string reference of this arrival; vote code of its base (agx_vote_code) and the read's strand
};

// Straight-line on purpose (no lane-varying branch, the two loads are issued for every lane with clamped indices): the number of
// loads in flight at any point of the sweep's loop is then static, and the compiler can wait for exactly the older buffer's loads
// (s_waitcnt vmcnt(2)) instead of draining everything.  Nothing loaded here is touched before apply().
AGX_HD void agx_arrival_fetch(const agx_sweep_args &A, const agx_dhit &d, agx_u32 X, bool enable, agx_pre &p) {
    const agx_arrival a = agx_decode_arrival(d, A.runs, X, A.k);
    const bool has = enable && a.has != 0;
    const bool rev = (d.flags & AGX_HF_AREV) != 0;
    p.has = has ? 1u : 0u;
    p.step1 = (has && a.has_succ && a.xs == X + 1) ? 1u : 0u;
    p.jump = (has && a.has_succ && a.xs != X + 1) ? 2u : 0u;
    p.type = a.type; p.p0 = a.p0; p.rev = rev ? 1u : 0u;
    p.s0 = d.a_slot;
    const agx_u32 stored = rev ? (agx_u32)d.len - 1u - a.q : a.q;                 // index of the arrival's base in the stored read
    p.s1 = (a.slen ? stored : 0u) | (a.slen << 16) | (rev ? 0x80000000u : 0u);
    p.mate = (has && a.p0 != AGX_NONE) ? 1u : 0u;
    p.h = A.cm_head[p.mate ? a.p0 : A.n_pos];                                      // no mate: the empty head
    p.cbyte = A.vcodes[(size_t)d.a_slot * A.stride + ((has && a.type == AGX_AT_K1) ? stored : 0u)];
}

// x -> x+1 edges of one hit, accumulated per position as a bit matrix: bit v*AGX_EM_W + w = variant v here has an edge to variant w of the
// next position.  Only kept for buckets of up to AGX_EM_W variants (the LDS sweep); larger buckets leave their edges to the edge passes.
#define AGX_EM_W 4u
// variant bit v of vm moved to bit v*AGX_EM_W
AGX_HD agx_u32 agx_edge_spread(agx_u32 vm) {
    return (vm & 1u) | ((vm & 2u) << (AGX_EM_W - 1u)) | ((vm & 4u) << (2u * AGX_EM_W - 2u)) | ((vm & 8u) << (3u * AGX_EM_W - 3u));
}
// sp = agx_edge_spread of the variants the hit touched here if it steps to position+1, else 0; vm_next = the variants it touched there.
// sp times the row (< 2^AGX_EM_W, so the partial products cannot overlap) puts a copy of the row at every touched variant.
AGX_HD void agx_edge_merge(agx_u32 &emask, agx_u32 sp, agx_u32 vm_next) { emask |= sp * (vm_next & ((1u << AGX_EM_W) - 1u)); }

// What an arrival that is NOT the straight-line case does to its position's bucket (shared by agx_node_sweep_lane and the device's lean pass-0 loop, agx_kernels.hip):
// the first arrival at a position with one candidate key stores variant 0 directly; anything else walks its candidate keys, X-major (AG:1369-1477), through
// agx_match_or_insert.  (cx_s, cx_n, cx0): the position's own conti-mers; (c0_s, c0_n, c0k): the mate position's (c0k NONE/NONE when there is no mate or it has none);
// vfield: the counter the base votes for (meaningful if votes).  Updates cnt / ok and the lane's copy of variant 0's mate-side key (v0_*: agx_node_sweep_lane); returns in
// vm the variants the arrival touched and in sp their agx_edge_spread form if it steps to position + 1.
AGX_HD void agx_arrival_slow(const agx_sweep_args &A, const agx_bucket &b, agx_u32 &cnt, bool &ok, agx_u32 cx_s, agx_u32 cx_n, agx_cmkey cx0,
                             agx_u32 p0, agx_u32 c0_s, agx_u32 c0_n, agx_cmkey c0k, agx_u32 s0, agx_u32 s1, agx_u32 is_k1, agx_u32 votes, agx_u32 vfield, agx_u32 step1,
                             agx_u32 &v0_ok, agx_u32 &v0_c0, agx_u32 &v0_o0, agx_u32 &v0_m, agx_u32 &vm, agx_u32 &sp) {
    if (AGX_SWEEP_FIRST && cnt == 0 && cx_n <= 1 && c0_n <= 1) {
        // The first arrival at a position with one candidate key — most of what leaves the fast path (7 % of the list entries on the
        // bench unit create a variant somewhere) — stores variant 0 without the general path's loops over candidates and variants.
        agx_b(b, 0, AGX_F_CID) = cx0.cid; agx_b(b, 0, AGX_F_COFF) = cx0.coff; agx_b(b, 0, AGX_F_CID0) = c0k.cid; agx_b(b, 0, AGX_F_COFF0) = c0k.coff;
        agx_b(b, 0, AGX_F_OFF0) = p0; agx_cnt_init(b, 0, is_k1);
        agx_b(b, 0, AGX_F_S0) = s0; agx_b(b, 0, AGX_F_S1) = s1;
        if (votes) agx_cnt_add<false>(b, 0, vfield, 1u);
        cnt = 1; vm = 1u; sp = step1;
        v0_ok = 1u; v0_c0 = c0k.cid; v0_o0 = c0k.coff; v0_m = p0;
        return;
    }
    if (ok) {
        const agx_u32 vf = votes ? vfield : (agx_u32)AGX_NF;
        const agx_u32 nx = cx_n ? cx_n : 1u, n0 = c0_n ? c0_n : 1u;
        for (agx_u32 ci = 0; ci < nx && ok; ci++) {                      // candidate keys, X-major (AG:1369-1477)
            agx_key key; key.off0 = p0;
            const agx_cmkey cx = cx_n ? (ci == 0 ? cx0 : A.cm[cx_s + ci]) : agx_cmkey{AGX_NONE, AGX_NONE};
            key.cid = cx.cid; key.coff = cx.coff;
            for (agx_u32 cj = 0; cj < n0; cj++) {
                const agx_cmkey c0 = c0_n ? (cj == 0 ? c0k : A.cm[c0_s + cj]) : agx_cmkey{AGX_NONE, AGX_NONE};
                key.cid0 = c0.cid; key.coff0 = c0.coff;
                const agx_u32 v = agx_match_or_insert(b, cnt, key, A.iv, is_k1 != 0, s0, s1);
                if (v == AGX_NONE) { ok = false; break; }
                vm |= 1u << (v & 31u);
                if (vf != AGX_NF) agx_cnt_add<false>(b, v, vf, 1u);
            }
        }
    }
    sp = step1 ? agx_edge_spread(vm) : 0u;
    if (cnt) { v0_ok = cx_n <= 1 ? 1u : 0u; v0_c0 = agx_b(b, 0, AGX_F_CID0); v0_o0 = agx_b(b, 0, AGX_F_COFF0); v0_m = agx_b(b, 0, AGX_F_OFF0); }
}

// The whole in-order sweep of one position.  get(i) returns the derived hit record of tile-list entry i (the kernels stage 64
// records at a time across the lanes of the wavefront and broadcast them; the test executor reads memory directly).
// Returns false if the bucket overflowed (the tile is then re-run with a larger bucket).
// exch(vm, sp) is called once per hit by every lane, in lockstep on the device: vm = the variants this hit's arrival touched at this
// position (bit v), sp = the same set in agx_edge_spread form if the hit steps from here to position+1, else 0.  It is how the sweep hands the x -> x+1 edges to agx_edge_merge:
// an event's k2 half at x+1 is the same hit's arrival there, so the edge set of x is the union over hits of (variants at x) x (variants
// at x+1) — known as soon as both lanes have applied the hit.
template <bool LDS_ADD, class GET, class EXCH>
AGX_HD bool agx_node_sweep_lane(const agx_sweep_args &A, agx_u32 tile, agx_u32 X, const agx_bucket &b, agx_u32 &cnt, agx_u32 &pflag, GET get, EXCH exch) {
    cnt = 0; pflag = 0;
    const bool live = X < A.n_pos;                       // lanes beyond the end still take part in the staging of hit records
    agx_u32 cx_s = 0, cx_n = 0; agx_cmkey cx0 = agx_cmkey{AGX_NONE, AGX_NONE};
    if (live) { const agx_cmhead h = A.cm_head[X]; cx_s = h.start; cx_n = h.n; cx0 = agx_cmkey{h.cid, h.coff}; }
    bool ok = true;
    const agx_u32 lo = A.tile_off[tile], hi = A.tile_off[tile + 1];
    // Branch-light on purpose: on wave64 every per-lane `if` costs exec-mask bookkeeping on the scalar unit, and the sweep runs
    // this body ~35 times per position; everything but the common case (several conti-mers, later variants, inserts) goes through agx_arrival_slow().
    // The common case — one candidate key, compatible with variant 0 — is straight-line code without a lane-varying branch and without
    // a read of the bucket.  A variant's key never changes once it is stored, so the lane keeps variant 0's mate-side key words in
    // registers from the moment it exists (v0_*); the position-side clause (AG:1375) needs no test at all there: with at most one
    // conti-mer at X every arrival carries the same (contigID, contigOffset) that variant 0 stored, and with several the arrival goes
    // through agx_arrival_slow() anyway (v0_ok = 0).  The verdict is integer arithmetic and the two counter updates are LDS adds of 0 or 1
    // (agx_bucket_add: ds_add_u32 in the LDS pass), so the loop's fast path never waits for LDS.  Only lanes that need another variant,
    // an insert or several candidate keys enter agx_arrival_slow().
    agx_u32 v0_ok = 0, v0_c0 = AGX_NONE, v0_o0 = AGX_NONE, v0_m = AGX_NONE;
    auto apply = [&](const agx_pre &p) {
        const agx_u32 has = p.has;                                                  // (a lane whose bucket overflowed keeps going: the tile is swept again)
        pflag |= p.step1 | p.jump;
        const agx_u32 is_k1 = p.type != AGX_AT_K2ONLY ? 1u : 0u, votes = p.type == AGX_AT_K1 ? 1u : 0u;
        const agx_u32 vfield = (p.cbyte >> (p.rev ? 4u : 0u)) & 15u;
        const agx_u32 c0_n = p.h.n;
        const agx_cmkey c0k = agx_cmkey{p.h.cid, p.h.coff};                        // NONE/NONE when the mate position has none (or there is no mate)
        const agx_u32 compat = agx_clause_ab(c0k.cid, c0k.coff, v0_c0, v0_o0, 2 * A.iv + AGX_EP25) & agx_clause_c(p.p0, v0_m, 2 * A.iv + AGX_EP25);
        const agx_u32 fast = has & v0_ok & compat & (agx_u32)(c0_n <= 1);
        const agx_u32 vote = fast & votes;
        agx_cnt_add<LDS_ADD>(b, 0, AGX_F_COV, fast & is_k1);
        agx_cnt_add<LDS_ADD>(b, 0, vote ? vfield : (agx_u32)AGX_F_COV, vote);
        agx_u32 vm = fast, sp = fast & p.step1;                                     // the straight-line case touches variant 0
        AGX_STAT(A, 0, true); AGX_STAT(A, 2, has != 0); AGX_STAT(A, 4, (has & (fast ^ 1u)) != 0);
        AGX_STAT(A, 6, (has & (fast ^ 1u)) != 0 && !(cnt == 0 && cx_n <= 1 && c0_n <= 1));
        AGX_STAT(A, 8, (has & (fast ^ 1u)) != 0 && cnt != 0 && cx_n <= 1 && c0_n <= 1 && v0_ok && !compat);      // one candidate, variant 0 exists and is not it
        AGX_STAT(A, 10, (has & (fast ^ 1u)) != 0 && (cx_n > 1 || c0_n > 1));                                        // several candidate keys
        if (has & (fast ^ 1u))
            agx_arrival_slow(A, b, cnt, ok, cx_s, cx_n, cx0, p.p0, p.h.start, c0_n, c0k, p.s0, p.s1, is_k1, votes, vfield, p.step1, v0_ok, v0_c0, v0_o0, v0_m, vm, sp);
        exch(vm, sp);
    };
    // software pipeline over two arrival buffers that are never copied (a register copy would have to wait for the loads): buffer
    // A serves the even hits of the list, B the odd ones; a buffer is refilled for the hit two places ahead right after it has been
    // applied, so its two per-lane loads have a whole apply + fetch of the other buffer to complete.  The derived hit record
    // (wave-uniform, a scalar load in the kernels) of the next refill is requested before the apply that precedes it.  Indices past
    // the end of the list read its last record again and are masked out.
    if (lo == hi) return true;
    auto rec = [&](agx_u32 i) { return get(i < hi ? i : hi - 1); };
    auto fetch = [&](const agx_dhit &d, bool valid, agx_pre &p) { AGX_STAT(A, 12, valid && (d.a_nruns | d.b_nruns) != 0); agx_arrival_fetch(A, d, X, valid && live, p); };
    agx_pre pa, pb;
    { const agx_dhit d0 = rec(lo); fetch(d0, true, pa); }
    { const agx_dhit d1 = rec(lo + 1); fetch(d1, lo + 1 < hi, pb); }
    for (agx_u32 i = lo; i < hi; i += 2) {
        const agx_dhit da = rec(i + 2);
        apply(pa); fetch(da, i + 2 < hi, pa);
        const agx_dhit db = rec(i + 3);
        apply(pb); fetch(db, i + 3 < hi, pb);
    }
    return ok;
}

// consensus base of a node (max, AG:1944-1952): 'X' if no votes; ties resolve A>C>G>T>N
AGX_HD char agx_consensus(agx_u32 a, agx_u32 c, agx_u32 g, agx_u32 t, agx_u32 n) {
    if (!(a | c | g | t | n)) return 'X';
    if (a >= c && a >= g && a >= t && a >= n) return 'A';
    if (c >= a && c >= g && c >= t && c >= n) return 'C';
    if (g >= a && g >= c && g >= t && g >= n) return 'G';
    if (t >= a && t >= c && t >= g && t >= n) return 'T';
    return 'N';
}

// write this position's bucket to the node table at node ids [base, base+cnt); prune (AG:1904-1918) and consensus fused in.
// edges != 0: the sweep's edge matrix is complete for this position (the next position lies in the same tile, both buckets stayed within
// AGX_EM_W variants): its x -> x+1 edges are written here — bn is the next position's bucket, [nbase, nbase+ncnt) its node ids — after the
// contig-consistency test between the two stored keys (AG:1602-1615), and bit 7 of pos_succ tells the edge passes so.
// Returns the position's side ids (surviving variants beyond the first); the caller turns them into side_pk / tile_side.
AGX_HD agx_u32 agx_node_write_lane(const agx_sweep_args &A, agx_u32 X, const agx_bucket &b, agx_u32 cnt, agx_u32 base, agx_u32 pflag,
                                   bool edges, agx_u32 emask, const agx_bucket &bn, agx_u32 nbase, agx_u32 ncnt) {
    if (X >= A.n_pos) return 0;
    A.node_start[X] = base; A.node_cnt[X] = (agx_u16)cnt; A.pos_succ[X] = (agx_u8)(pflag | (edges ? 0x80u : 0u));
    agx_u32 alive = 0;
    for (agx_u32 v = 0; v < cnt; v++) {
        const agx_u32 id = base + v;
        const agx_u32 cid = agx_b(b, v, AGX_F_CID), coff = agx_b(b, v, AGX_F_COFF), cov = agx_cnt_get(b, v, AGX_F_COV);
        const agx_u32 cid0 = agx_b(b, v, AGX_F_CID0), coff0 = agx_b(b, v, AGX_F_COFF0);
        A.nk_cid[id] = cid; A.nk_coff[id] = coff; A.nk_cid0[id] = cid0; A.nk_coff0[id] = coff0;
        A.nk_off0[id] = agx_b(b, v, AGX_F_OFF0);      // (r06: no per-node position array — a node's position is its walk id's: the id itself, or side_xpos — 4 bytes per node less to write)
        const agx_u32 va = agx_cnt_get(b, v, AGX_F_A), vc = agx_cnt_get(b, v, AGX_F_C), vg = agx_cnt_get(b, v, AGX_F_G), vt = agx_cnt_get(b, v, AGX_F_T), vn = agx_cnt_get(b, v, AGX_F_N);
        A.n_base[id] = (agx_u8)agx_consensus(va, vc, vg, vt, vn);
        agx_u8 fl = 0;
        if (cid == AGX_NONE && (int)cov < A.coverage) fl |= AGX_NF_DEAD; else alive++;
        if (coff != AGX_NONE) fl |= AGX_NF_CONTIG;
        A.n_flags[id] = fl;
        agx_sref s; s.slot = agx_b(b, v, AGX_F_S0); s.qlen = agx_b(b, v, AGX_F_S1); A.n_sref[id] = s;
        agx_u32 slot[AGX_MAXE]; agx_u32 k = 0;
        for (agx_u32 e = 0; e < AGX_MAXE; e++) slot[e] = AGX_NONE;
        if (edges && v < AGX_EM_W)
            for (agx_u32 w = 0; w < AGX_EM_W && w < ncnt; w++) {
                if (!((emask >> (v * AGX_EM_W + w)) & 1u)) continue;
                const agx_u32 ok = agx_clause_ab(agx_b(bn, w, AGX_F_CID), agx_b(bn, w, AGX_F_COFF), cid, coff, AGX_EP25) &
                                   agx_clause_ab(agx_b(bn, w, AGX_F_CID0), agx_b(bn, w, AGX_F_COFF0), cid0, coff0, 2 * A.iv + AGX_EP25);       // agx_edge_allowed
                if (ok) slot[k++] = nbase + w;
            }
        for (agx_u32 e = 0; e < AGX_MAXE; e++) A.n_next[(size_t)id * AGX_MAXE + e] = slot[e];
        if (A.n_counts) { int *c = A.n_counts + (size_t)id * 6; c[0] = (int)cov; c[1] = (int)va; c[2] = (int)vc; c[3] = (int)vg; c[4] = (int)vt; c[5] = (int)vn; }
    }
    return alive ? alive - 1 : 0;                   // walk ids: the first surviving variant takes the position's main id, the others go to the side block
}
static_assert(63u * (AGX_MAXV_HUGE - 1u) < 65536u && AGX_MAXV_HUGE <= 65535u, "side ids of a tile's first 63 positions must fit 16 bits");
AGX_HD agx_u32 agx_side_pack(agx_u32 before_in_tile, agx_u32 here) { return before_in_tile | (here << 16); }

// ---- edge sweep (AG:1589-1623) ---------------------------------------------------------------------------------

struct agx_edge_ovf { agx_u32 src, dst; };

// first compatible variant of the FINAL bucket at position x for key; NONE if x has no compatible variant (cannot happen for a real arrival)
AGX_HD agx_u32 agx_resolve(const agx_sweep_args &A, agx_u32 x, const agx_key &k) {
    const agx_u32 s = A.node_start[x], n = A.node_cnt[x];
    for (agx_u32 v = 0; v < n; v++) {
        const agx_u32 id = s + v;
        if (agx_clause_ab(k.cid, k.coff, A.nk_cid[id], A.nk_coff[id], AGX_EP25) &
            agx_clause_ab(k.cid0, k.coff0, A.nk_cid0[id], A.nk_coff0[id], 2 * A.iv + AGX_EP25) &
            agx_clause_c(k.off0, A.nk_off0[id], 2 * A.iv + AGX_EP25)) return id;
    }
    return AGX_NONE;
}

// contig-consistency between the STORED keys of an edge's two ends (AG:1602-1615)
AGX_HD bool agx_edge_allowed(const agx_sweep_args &A, agx_u32 src, agx_u32 dst) {
    return (agx_clause_ab(A.nk_cid[dst], A.nk_coff[dst], A.nk_cid[src], A.nk_coff[src], AGX_EP25) &
            agx_clause_ab(A.nk_cid0[dst], A.nk_coff0[dst], A.nk_cid0[src], A.nk_coff0[src], 2 * A.iv + AGX_EP25)) != 0;
}

// Edge build, pass A (lanes = positions).  Where a position holds ONE variant every arrival resolved to it, so no candidate keys and
// no compatibility tests are needed on the source side: if x+1 holds one variant too, the edge x -> x+1 exists iff the node sweep saw
// an arrival stepping to x+1 (pos_succ) and the contig-consistency predicate of the two stored keys holds — no per-hit work at all.
// Positions with several variants, or whose neighbour x+1 has several, return true ("slow"): pass B re-resolves every hit for them.
// Only this lane writes n_next of x's node in this pass, so plain stores suffice.  Steps that do not go to x+1 are pass J's.
// Since the node sweep writes the x -> x+1 edges itself wherever x+1 lies in the same tile (agx_edge_merge), this pass is left with
// the last position of every tile and with tiles whose buckets outgrew LDS.
AGX_HD bool agx_edge_fast_lane(const agx_sweep_args &A, agx_u32 X, agx_u32 own_start, agx_u32 own_cnt, agx_u32 nb_start, agx_u32 nb_cnt) {
    const bool live = X < A.n_pos && own_cnt != 0;
    if (X >= A.n_pos) X = 0;
    const agx_u32 ps = A.pos_succ[X];
    // The node sweep already wrote this position's x -> x+1 edges (bit 7): all that can be missing are steps to other positions, which
    // pass J adds for single-variant sources; a multi-variant source with such a step (bit 1) goes through pass B once more.
    if (ps & 0x80u) return live && own_cnt >= 2 && (ps & 2u);
    const bool fast = live && own_cnt == 1;
    const bool slow = live && (own_cnt >= 2 || nb_cnt >= 2);
    if (fast) {
        agx_u32 s0 = AGX_NONE;
        if (nb_cnt == 1 && (ps & 1u) && agx_edge_allowed(A, own_start, nb_start)) s0 = nb_start;
        agx_u32 *slots = A.n_next + (size_t)own_start * AGX_MAXE;
        slots[0] = s0; slots[1] = AGX_NONE; slots[2] = AGX_NONE; slots[3] = AGX_NONE;
    }
    return slow;
}

// Edge build, pass J (lanes = hits): steps that do not go to x+1 (read deletions, AG:1822-1838).  They only exist where a run of the
// hit's a mate ends, so each hit with several runs (one hit in ten) visits its run ends; sources with one variant are handled here
// (set insert through INS, after pass A's plain stores), sources with several variants belong to pass B like all their other edges.
template <class INS>
AGX_HD void agx_edge_jump_hit(const agx_sweep_args &A, const agx_dhit &d, INS ins) {
    if ((d.flags & AGX_HF_SKIP) || d.a_nruns < 2) return;
    for (agx_u32 i = 0; i + 1 < d.a_nruns; i++) {
        const agx_run r = A.runs[d.a_runs + i];
        if (r.n == 0) continue;
        const agx_u32 X = r.t + r.n - 1;
        if (X >= A.n_pos || A.node_cnt[X] != 1) continue;
        const agx_arrival a = agx_decode_arrival(d, A.runs, X, A.k);
        if (!a.has || !a.has_succ || a.xs >= A.n_pos || a.xs == X + 1) continue;
        const agx_u32 src = A.node_start[X];
        const agx_u32 sx_s = A.cm_start[a.xs], sx_n = A.cm_start[a.xs + 1] - sx_s;
        agx_for_candidates(A, sx_s, sx_n, a.p0s, [&](const agx_key &k2) {
            const agx_u32 dst = agx_resolve(A, a.xs, k2);
            if (dst != AGX_NONE && agx_edge_allowed(A, src, dst)) ins(src, dst);
            return true;
        });
    }
}

// Edge build, pass B (lanes = hits): every edge out of a slow position x that ONE hit contributes (AG:1589-1623).  Edge order is
// irrelevant to the walk (it only counts unvisited successors, AG:2020-2033), so the hits of x's tile are resolved in parallel and
// INS(src, dst) performs a set insert into src's slots (compare-and-swap on the device).
template <class INS>
AGX_HD void agx_edge_slow_hit(const agx_sweep_args &A, agx_u32 X, const agx_dhit &d, INS ins) {
    const agx_arrival a = agx_decode_arrival(d, A.runs, X, A.k);
    if (!a.has || !a.has_succ || a.xs >= A.n_pos) return;
    // X is wave-uniform on the device (one wavefront per slow position), and so is X+1 — the usual successor.  Keeping the two cases
    // apart lets every bucket/conti-mer load of the common case go through the scalar cache instead of 64 per-lane gathers.
    const agx_u32 cx_s = A.cm_start[X], cx_n = A.cm_start[X + 1] - cx_s;
    const bool step1 = a.xs == X + 1;
    const agx_u32 xs = step1 ? X + 1 : a.xs;
    agx_u32 sx_s, sx_n;
    if (step1) { sx_s = A.cm_start[X + 1]; sx_n = A.cm_start[X + 2] - sx_s; } else { sx_s = A.cm_start[xs]; sx_n = A.cm_start[xs + 1] - sx_s; }
    agx_for_candidates(A, cx_s, cx_n, a.p0, [&](const agx_key &k1) {
        const agx_u32 src = agx_resolve(A, X, k1);
        if (src == AGX_NONE) return true;
        agx_for_candidates(A, sx_s, sx_n, a.p0s, [&](const agx_key &k2) {
            const agx_u32 dst = step1 ? agx_resolve(A, X + 1, k2) : agx_resolve(A, xs, k2);
            if (dst != AGX_NONE && agx_edge_allowed(A, src, dst)) ins(src, dst);
            return true;
        });
        return true;
    });
}

// Pass B, common case.  Everything about the slow position X itself is wave-uniform on the device: its bucket, the bucket of X+1 (the
// usual successor), their conti-mers.  When both buckets are small the stored keys are loaded ONCE per position (scalar loads, SGPRs)
// and every hit resolves both ends of its edge against them in registers — no per-hit walk through the node table.  A hit then only
// names one of at most 16 (source variant, target variant) pairs; the pairs a position produces are OR-ed over its hits and inserted
// once each.  Hits whose successor is not X+1, or that have several candidate keys, take agx_edge_slow_hit as before.
#define AGX_SLOW_V 4u
struct agx_slow_ctx {
    agx_u32 reg;                                   // 1: the register path applies to this position
    agx_u32 s, n, s1, n1;                          // buckets of X and X+1 (node ids s..s+n-1, s1..s1+n1-1)
    agx_cmkey cx, sx;                              // the conti-mer of X / of X+1 (NONE/NONE if none)
    agx_key kx[AGX_SLOW_V], kx1[AGX_SLOW_V];       // stored keys of the variants
    agx_u32 allowed;                               // bit vs*AGX_SLOW_V+vd: edge s+vs -> s1+vd passes agx_edge_allowed
};

AGX_HD agx_u32 agx_key_compatible(const agx_key &k, const agx_key &st, int iv) {
    return agx_clause_ab(k.cid, k.coff, st.cid, st.coff, AGX_EP25) & agx_clause_ab(k.cid0, k.coff0, st.cid0, st.coff0, 2 * iv + AGX_EP25) &
           agx_clause_c(k.off0, st.off0, 2 * iv + AGX_EP25);
}

AGX_HD void agx_edge_slow_ctx(const agx_sweep_args &A, agx_u32 X, agx_slow_ctx &c) {
    c.s = agx_uload(A.node_start, X); c.n = A.node_cnt[X]; c.s1 = 0; c.n1 = 0; c.allowed = 0;
    c.cx = agx_cmkey{AGX_NONE, AGX_NONE}; c.sx = c.cx;
    const bool has1 = X + 1 < A.n_pos;
    if (has1) { c.s1 = agx_uload(A.node_start, X + 1); c.n1 = A.node_cnt[X + 1]; }
    const agx_u32 *hw = reinterpret_cast<const agx_u32 *>(A.cm_head);            // {cid, coff, n, start} per position
    const agx_u32 hx_n = agx_uload(hw, 4 * (size_t)X + 2), hx1_n = has1 ? agx_uload(hw, 4 * (size_t)X + 6) : 0u;
    c.reg = (has1 && c.n <= AGX_SLOW_V && c.n1 <= AGX_SLOW_V && hx_n <= 1 && hx1_n <= 1) ? 1u : 0u;
    if (!c.reg) return;
    c.cx = agx_cmkey{agx_uload(hw, 4 * (size_t)X), agx_uload(hw, 4 * (size_t)X + 1)};
    c.sx = agx_cmkey{agx_uload(hw, 4 * (size_t)X + 4), agx_uload(hw, 4 * (size_t)X + 5)};
    const agx_key none = agx_key{AGX_NONE, AGX_NONE, AGX_NONE, AGX_NONE, AGX_NONE};
    for (agx_u32 v = 0; v < AGX_SLOW_V; v++) {
        c.kx[v] = none; c.kx1[v] = none;
        // the device reads all AGX_SLOW_V rows of both buckets in one batch of scalar loads (rows past a bucket's end are other nodes'
        // keys inside the pool, which is allocated with that much slack, and are never looked at)
#if defined(__HIP_DEVICE_COMPILE__)
        const bool in0 = true, in1 = true;
#else
        const bool in0 = v < c.n, in1 = v < c.n1;
#endif
        if (in0) { const size_t id = (size_t)c.s + v; c.kx[v] = agx_key{agx_uload(A.nk_cid, id), agx_uload(A.nk_coff, id), agx_uload(A.nk_cid0, id), agx_uload(A.nk_coff0, id), agx_uload(A.nk_off0, id)}; }
        if (in1) { const size_t id = (size_t)c.s1 + v; c.kx1[v] = agx_key{agx_uload(A.nk_cid, id), agx_uload(A.nk_coff, id), agx_uload(A.nk_cid0, id), agx_uload(A.nk_coff0, id), agx_uload(A.nk_off0, id)}; }
    }
    for (agx_u32 vs = 0; vs < AGX_SLOW_V; vs++) for (agx_u32 vd = 0; vd < AGX_SLOW_V; vd++) {
        if (vs >= c.n || vd >= c.n1) continue;
        const agx_u32 ok = agx_clause_ab(c.kx1[vd].cid, c.kx1[vd].coff, c.kx[vs].cid, c.kx[vs].coff, AGX_EP25) &
                           agx_clause_ab(c.kx1[vd].cid0, c.kx1[vd].coff0, c.kx[vs].cid0, c.kx[vs].coff0, 2 * A.iv + AGX_EP25);      // agx_edge_allowed
        c.allowed |= ok << (vs * AGX_SLOW_V + vd);
    }
}

// one hit of a slow position: returns the (source, target) pair it contributes as a bit (0: none), or — for hits off the register
// path — resolves through the node table and inserts through ins() right away
template <class INS>
AGX_HD agx_u32 agx_edge_slow_pair(const agx_sweep_args &A, const agx_slow_ctx &c, agx_u32 X, const agx_dhit &d, bool enable, INS ins) {
    const agx_arrival a = agx_decode_arrival(d, A.runs, X, A.k);
    const bool act = enable && a.has && a.has_succ && a.xs < A.n_pos;
    const bool mate = act && a.p0 != AGX_NONE, mates = act && a.p0s != AGX_NONE;
    const agx_cmhead h0 = A.cm_head[mate ? a.p0 : 0u], h0s = A.cm_head[mates ? a.p0s : 0u];                 // clamped, unconditional loads
    const bool quick = act && c.reg && a.xs == X + 1 && (!mate || h0.n <= 1) && (!mates || h0s.n <= 1);
    if (act && !quick) agx_edge_slow_hit(A, X, d, ins);
    const agx_key k1 = agx_key{c.cx.cid, c.cx.coff, mate ? h0.cid : AGX_NONE, mate ? h0.coff : AGX_NONE, a.p0};
    const agx_key k2 = agx_key{c.sx.cid, c.sx.coff, mates ? h0s.cid : AGX_NONE, mates ? h0s.coff : AGX_NONE, a.p0s};
    agx_u32 vs = AGX_NONE, vd = AGX_NONE;
    for (agx_u32 v = 0; v < AGX_SLOW_V; v++) {                                     // first compatible variant (agx_resolve), select-only
        const bool m1 = v < c.n && agx_key_compatible(k1, c.kx[v], A.iv) != 0, m2 = v < c.n1 && agx_key_compatible(k2, c.kx1[v], A.iv) != 0;
        vs = (vs == AGX_NONE && m1) ? v : vs; vd = (vd == AGX_NONE && m2) ? v : vd;
    }
    const bool hit = quick && vs != AGX_NONE && vd != AGX_NONE;
    const agx_u32 bit = 1u << ((hit ? vs : 0u) * AGX_SLOW_V + (hit ? vd : 0u));
    return (hit && (c.allowed & bit)) ? bit : 0u;
}

// ---- walk preparation: alive-node renumbering and forced-run flags -------------------------------------------------------
// The path walk (AG:1954-2204) only ever stands on nodes that survived the coverage prune.  After the edge sweep the
// surviving ("alive") nodes get walk ids ("aid") laid out so that the main strand of the graph is contiguous:
//     aid = X                          for the FIRST alive variant of position X          (main block, one slot per position)
//     aid = n_pos + side ids of all earlier positions + j  for the (j+1)-th further alive variant of X   (side block, position-major;
//           = tile_side_start[tile of X] + the in-tile prefix the node sweep left in side_pk[X])
// so scanning "every position, every variant, if untraversed" (AG:1972-1978) is: slot X, then X's side range.
// Out-edges are rewritten in aids with pruned targets dropped (a pruned node is `traversed` from the start, AG:1915, and
// can never count as a live successor, AG:2027).  Every node gets a `cont` byte: 1 iff its only alive successor is aid+1.
// Standing on such a node the walk has exactly one choice — step to aid+1 if that node is still unvisited, else stop
// (AG:2020-2046) — so the host replays a run of cont==1 nodes with memchr/memcpy (the first already-visited node ends the
// run) and only evaluates real branch points, jumps and dead ends.

// per-node record the walk reads at branch points, record starts and record ends: one 32-byte line instead of four arrays
struct agx_walknode { agx_u32 next[AGX_MAXE]; agx_u32 off0, xpos; agx_sref sref; };

// what a walk does when it leaves the k-mer graph at a position (AG:2047-2057): exactly one conti-mer there, with a next -> append
// chain_str[str_off, str_off+len) and land on end_pos; len == 0: no hop possible.  Kept per position on the host (Threads::hop); the
// engine gathers the entries of the special ids' positions next to their records when it downloads the sparse table.
struct agx_hop { agx_u32 str_off, len, end_pos; };

// a_meta bits: forced step to id+1; contigOffset != -1 (AG:2004); (main ids) the position has further alive variants in the side
// block; (main ids) the position holds at least one variant, pruned or not (scaffold gap rule, AG:2428); no node at this id
enum { AGX_WM_CONT = 1, AGX_WM_CONTIG = 2, AGX_WM_SIDE = 4, AGX_WM_ANY = 8, AGX_WM_ABSENT = 128 };

// The host walk reads a node's 32-byte record only where a walk can start, stop, branch or land: everything inside a forced run is
// covered by the meta and base bytes.  The device therefore hands over a SPARSE record table: the records of the "special" ids in id
// order, a bitmap of which ids are special and a per-64-ids rank (popcount prefix) to find a record.  Special ids are
//   - every side id (their position is read from the record),
//   - every id without the cont bit (branch points, dead ends) and every id whose predecessor id lacks it (run heads: the only
//     places the position scan of AG:1972-1978 can start a walk, since a cont predecessor drags its successor along),
//   - every edge target of an id without the cont bit, and the id just before such a target (a run stops in front of a node that a
//     jump has already visited),
//   - every main id of a position where a conti-mer chain ends (the hop back onto the k-mer graph, AG:2093-2136, reads the node there).
// The one place a walk stands anywhere else is after the reference's +1000 position skip inside long records (AG:2194-2202), which
// can start a walk in the middle of a forced run; the host then fetches that record from the full table that stays on the device.
struct agx_compact_args {
    // node table, old ids
    const agx_u32 *node_start; const agx_u16 *node_cnt; const agx_u8 *n_flags; const agx_u8 *n_base;
    const agx_u32 *nk_off0; const agx_sref *n_sref; const agx_u32 *n_next; const char *ref;
    agx_u32 n_pos;
    const agx_u32 *side_pk;        // [n_pos] agx_side_pack
    const agx_u32 *tile_side_start; // [n_tiles+1] exclusive scan of tile_side: side ids of all earlier tiles
    agx_u32 *aid_of;               // [pool] walk id or NONE
    // outputs indexed by aid, [n_pos + n_side]
    char *a_str; agx_u8 *a_meta;   // a_meta: AGX_WM_* bits
    agx_u32 *a_nid;                // node of every walk id (NONE: absent main id): node records are built from the node table where one is wanted (agx_walk_record)
    const agx_edge_ovf *ovf; agx_u32 n_ovf; agx_edge_ovf *a_ovf;   // overflow edges rewritten in aids (edges touching pruned nodes become NONE/NONE)
    agx_u8 *a_mark;                // [n_ids+1] zeroed before; 1 = edge target of a non-cont id, or main id of a chain-end position
    agx_u32 *side_xpos;            // [n_side] position of every side id (sorted: the side block is position-major)
    // sparse record table
    agx_u32 n_ids, sparse_min;     // sparse_min (test hook): only the side ids are special, every other record comes through the fetch path
    unsigned long long *sp_bits; agx_u32 *sp_cnt; const agx_u32 *sp_rank; agx_walknode *sp_node;
    agx_u32 sp_cap;                        // records the sparse table can hold (the host grows it and repeats the build if there are more)
    const agx_cmseg *segs; agx_u32 n_seg0; const agx_u32 *cm_start; agx_hop *sp_hop;   // device only: hop entries of the special ids' positions, from the runs
    const agx_u32 *seg_index;   // device only: [n_pos / AGX_SEG_INDEX + 2] last rank-0 run that starts at or before position i * AGX_SEG_INDEX (0 if none does): where the search for a position's run begins
    const agx_u32 *abort;          // device only: the node sweeps' status word (non-zero: the node table is incomplete, the kernels do nothing)
};

// per position, after the scan: walk ids of its nodes; main slots without an alive node are marked absent (= visited from the start)
AGX_HD void agx_assign_aid_pos(const agx_compact_args &A, agx_u32 X) {
    if (X >= A.n_pos) return;
    const agx_u32 s = A.node_start[X], n = A.node_cnt[X]; agx_u32 side = A.n_pos + A.tile_side_start[X / AGX_TILE] + (A.side_pk[X] & 0xFFFFu); bool first = true;
    for (agx_u32 v = 0; v < n; v++) {
        if (A.n_flags[s + v] & AGX_NF_DEAD) { A.aid_of[s + v] = AGX_NONE; continue; }
        if (first) { A.aid_of[s + v] = X; first = false; } else A.aid_of[s + v] = side++;
    }
    if (first) { A.a_meta[X] = (agx_u8)(AGX_WM_ABSENT | (n ? AGX_WM_ANY : 0)); A.a_str[X] = 'N'; A.a_nid[X] = AGX_NONE; }
}

// per old node of position x: string character, flags and marks of its walk id
AGX_HD void agx_emit_alive_node(const agx_compact_args &A, agx_u32 v, agx_u32 x) {
    const agx_u32 a = A.aid_of[v];
    if (a == AGX_NONE) return;
    const char c = (char)A.n_base[v];
    A.a_str[a] = c != 'X' ? c : A.ref[x];                 // consensus, else the reference base (AG:1997-2001)
    A.a_nid[a] = v;
    agx_u32 next[AGX_MAXE]; agx_u32 k = 0;
    for (agx_u32 e = 0; e < AGX_MAXE; e++) {
        const agx_u32 t = A.n_next[(size_t)v * AGX_MAXE + e];
        if (t == AGX_NONE) break;
        const agx_u32 ta = A.aid_of[t];
        if (ta != AGX_NONE) next[k++] = ta;
    }
    // a node whose edges spilled to the overflow list keeps all four slots... unless some pointed at pruned nodes: mark it
    // by never being `cont`; the host consults the overflow list for every node it finds there
    const bool cont = k == 1 && !(A.n_flags[v] & AGX_NF_EOVF) && next[0] == a + 1;
    agx_u8 m = (agx_u8)((cont ? AGX_WM_CONT : 0) | ((A.n_flags[v] & AGX_NF_CONTIG) ? AGX_WM_CONTIG : 0));
    if (a < A.n_pos) m |= (agx_u8)(AGX_WM_ANY | ((A.side_pk[x] >> 16) ? AGX_WM_SIDE : 0));
    else A.side_xpos[a - A.n_pos] = x;
    A.a_meta[a] = m;
    if (!cont) for (agx_u32 e = 0; e < k; e++) A.a_mark[next[e]] = 1;      // racing stores of the same value
}

// The node record of walk id a (successor ids in slot order with the pruned ones dropped, mate offset, position, k-mer string
// reference), from the node table.  Only the special ids' records travel to the host with the graph (agx_k_special_emit); the walk asks
// for any other one by id (the +1000 skip, AG:2194-2202) and gets it from the same function.
AGX_HD agx_walknode agx_walk_record(const agx_compact_args &A, agx_u32 a) {
    agx_walknode w; for (agx_u32 e = 0; e < AGX_MAXE; e++) w.next[e] = AGX_NONE;
    const agx_u32 v = A.a_nid[a];
    if (v == AGX_NONE) { w.off0 = AGX_NONE; w.xpos = a; w.sref = agx_sref{0, 0}; return w; }       // a main id without a surviving node
    w.off0 = A.nk_off0[v]; w.xpos = a < A.n_pos ? a : A.side_xpos[a - A.n_pos]; w.sref = A.n_sref[v];      // (a walk id's position: main ids are positions, the side block is position-major)
    agx_u32 k = 0;
    for (agx_u32 e = 0; e < AGX_MAXE; e++) {
        const agx_u32 t = A.n_next[(size_t)v * AGX_MAXE + e];
        if (t == AGX_NONE) break;
        const agx_u32 ta = A.aid_of[t];
        if (ta != AGX_NONE) w.next[k++] = ta;
    }
    return w;
}

// per position: its nodes (the node table is only ever entered through node_start / node_cnt: the pool it lives in has unused slots)
AGX_HD void agx_emit_alive_pos(const agx_compact_args &A, agx_u32 X) {
    if (X >= A.n_pos) return;
    const agx_u32 s = A.node_start[X], n = A.node_cnt[X];
    for (agx_u32 v = 0; v < n; v++) agx_emit_alive_node(A, s + v, X);
}

AGX_HD void agx_emit_alive_ovf(const agx_compact_args &A, agx_u32 i) {
    if (i >= A.n_ovf) return;
    const agx_u32 s = A.aid_of[A.ovf[i].src], d = A.aid_of[A.ovf[i].dst];
    A.a_ovf[i] = (s == AGX_NONE || d == AGX_NONE) ? agx_edge_ovf{AGX_NONE, AGX_NONE} : agx_edge_ovf{s, d};
    if (s != AGX_NONE && d != AGX_NONE) A.a_mark[d] = 1;      // the source spilled, so it is never cont
}

// is walk id a special (does its record go into the sparse table)?  Runs after every a_meta / a_mark store of the unit.
AGX_HD bool agx_special_id(const agx_compact_args &A, agx_u32 a) {
    if (a >= A.n_ids) return false;
    // the four bytes are loaded together (a chain of `||` made them four dependent loads: 0.42 ms for a 30 Mb unit whose kernel moved 56 MB)
    const agx_u8 m = A.a_meta[a], mp = a ? A.a_meta[a - 1] : (agx_u8)0, k0 = A.a_mark[a], k1 = A.a_mark[a + 1];
    if (m & AGX_WM_ABSENT) return false;
    if (a >= A.n_pos) return true;
    if (A.sparse_min) return false;
    return (!(m & AGX_WM_CONT)) | (a == 0) | (!(mp & AGX_WM_CONT)) | (k0 != 0) | (k1 != 0);
}
