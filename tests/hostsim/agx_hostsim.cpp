// agx_hostsim — TEST-ONLY serial executor of the engine's kernels.
//
// Runs exactly the per-lane functions of aligngraph_amd/csrc/agx_core.h that the gfx950 kernels run
// (hit_prep -> tile binning -> node sweep -> edge sweep), one "lane" after another on the CPU, followed by
// the product's own host walk.  It exists so that `pytest -m "not gpu"` can check the re-formulated algorithm
// (merged arrivals, in-order tile sweeps, edges resolved against final buckets) against the oracle in a
// container without a GPU.  It is never linked into libagx.so and the product API cannot reach it.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../aligngraph_amd/csrc/agx_host.h"

using namespace agx;

namespace {

struct SimGraph {
    std::vector<agx_u32> node_start; std::vector<agx_u16> node_cnt; std::vector<agx_u8> pos_succ;
    std::vector<agx_u32> cid, coff, cid0, coff0, off0, xpos, next;
    std::vector<agx_u8> base, flags; std::vector<agx_sref> sref; std::vector<int> counts;
    std::vector<agx_edge_ovf> ovf;
    agx_u32 n_nodes = 0;
    // alive-compacted view (agx_core.h "walk preparation")
    agx_u32 n_ids = 0;
    std::vector<agx_u32> side_pk, tile_side, tile_side_start, aid_of; std::string a_str;
    std::vector<agx_u8> a_meta, a_mark; std::vector<agx_walknode> sp_node; std::vector<agx_u32> a_nid; agx_compact_args walk_args; std::vector<agx_hop> sp_hop; std::vector<agx_edge_ovf> a_ovf;
    std::vector<agx_u32> side_xpos, sp_cnt, sp_rank; std::vector<unsigned long long> sp_bits; agx_u32 n_special = 0;
    void reserve(size_t cap) {
        cid.resize(cap); coff.resize(cap); cid0.resize(cap); coff0.resize(cap); off0.resize(cap); xpos.resize(cap); next.resize(cap * AGX_MAXE);
        base.resize(cap); flags.resize(cap); sref.resize(cap); counts.resize(cap * 6);
    }
};

void simulate(const Threads &T, const Pairs &P, agx_u32 k, int iv, int coverage, agx_u32 maxv_first, SimGraph &S, int &n_big_tiles) {
    const agx_u32 n_pos = (agx_u32)T.ref.size();
    const agx_u32 n_tiles = (n_pos + AGX_TILE - 1) / AGX_TILE;
    // hit_prep
    std::vector<agx_dhit> dh(P.hits.size());
    for (agx_u32 h = 0; h < P.hits.size(); h++)
    {   const bool swap = agx_hit_left_is_mate2(P.hits[h], P.runs.data(), k);      // (the engine decides this when it stages the hits, and uploads only the a mates' bases)
        if (agx_hit_prep(P.hits[h], agx_hit_dup(P.hits.data(), P.runs.data(), h), swap, P.hits[h].slot1 + (swap ? 1u : 0u), P.runs.data(), k, dh[h]) != 0) throw Error{E_ALIGNMENT, "BOWTIE ALIGNMENT ERROR"}; }
    // binning: (tile, hit) pairs in (tile, hit) order
    std::vector<std::vector<agx_u32> > lists(n_tiles);
    for (agx_u32 h = 0; h < dh.size(); h++) {
        if (dh[h].flags & AGX_HF_SKIP) continue;
        if (dh[h].x_hi >= n_pos) throw Error{E_FORMAT, "alignment beyond the end of the unit sequence"};
        for (agx_u32 t = dh[h].x_lo / AGX_TILE; t <= dh[h].x_hi / AGX_TILE; t++) lists[t].push_back(h);
    }
    std::vector<agx_u32> tile_off(n_tiles + 1, 0), tile_hits;
    size_t lean_kinds[4] = {0, 0, 0, 0};
    std::vector<agx_dhit> tile_recs;                       // what agx_k_tile_sort writes: per list entry the hit's record for THAT tile (agx_tile_record)
    for (agx_u32 t = 0; t < n_tiles; t++) {
        tile_off[t] = (agx_u32)tile_hits.size(); tile_hits.insert(tile_hits.end(), lists[t].begin(), lists[t].end());
        for (agx_u32 h : lists[t]) {
            const agx_dhit piece = agx_tile_record(dh[h], P.runs.data(), t, k);
            // a piece must decode, on every position of its tile, to exactly what the hit's full record decodes to
            for (agx_u32 X = t * AGX_TILE; X < (t + 1) * AGX_TILE && X < n_pos; X++) {
                const agx_arrival a = agx_decode_arrival(dh[h], P.runs.data(), X, k), b = agx_decode_arrival(piece, P.runs.data(), X, k);
                const bool same = a.has == b.has && (!a.has || (a.type == b.type && a.q == b.q && a.slen == b.slen && a.p0 == b.p0 && a.has_succ == b.has_succ &&
                                  (!a.has_succ || (a.xs == b.xs && a.p0s == b.p0s))));
                if (!same) throw Error{E_ARG, "a tile record decodes differently from the hit's record"};
            }
            tile_recs.push_back(piece);
            // the lean record of the entry (what pass 0 of the device's node sweep reads): wherever it is not kind GENERAL it must give the same arrival on every lane
            const agx_lrec lr = agx_lean_make(dh[h], P.runs.data(), t, k, h);
            lean_kinds[lr.geo >> 30]++;
            if (lr.slot != dh[h].a_slot || lr.hit != h || lr.lenjs != ((agx_u32)dh[h].len | ((agx_u32)dh[h].jstar << 16)) || (((lr.geo & AGX_LF_AREV) != 0) != ((dh[h].flags & AGX_HF_AREV) != 0)))
                throw Error{E_ARG, "a lean tile record names another read"};
            if ((lr.geo >> 30) == AGX_LK_ONE && (lr.qoff2 != ((dh[h].flags & AGX_HF_AREV) ? (agx_u32)dh[h].len - lr.qoff1 : lr.qoff1) || lr.boff2 != (agx_u32)dh[h].jstar - lr.qoff1 || (lr.geo & (AGX_LF_BN1 | AGX_LF_JUMP1))))
                throw Error{E_ARG, "a one-piece lean record's digested fields are not what its offsets say"};
            if ((lr.geo >> 30) != AGX_LK_GENERAL) {
                if (lr.slot != dh[h].a_slot || lr.hit != h || lr.lenjs != ((agx_u32)dh[h].len | ((agx_u32)dh[h].jstar << 16)) || (((lr.geo & AGX_LF_AREV) != 0) != ((dh[h].flags & AGX_HF_AREV) != 0)))
                    throw Error{E_ARG, "a lean tile record names another read"};
                for (agx_u32 lane = 0; lane < AGX_TILE; lane++) {
                    const agx_u32 X = t * AGX_TILE + lane;
                    const agx_arrival a = agx_decode_arrival(dh[h], P.runs.data(), X, k); const agx_larr l = agx_lean_decode(lr, lane);
                    const bool a_has = a.has != 0 && X < n_pos;
                    bool same = a_has == (l.has != 0);
                    if (same && a_has) same = a.type != AGX_AT_CHAIN && (a.type == AGX_AT_K2ONLY) == (l.last != 0) && a.q == l.q && a.p0 == l.p0 &&
                                              (a.has_succ != 0) == (l.last == 0) && (!a.has_succ || ((a.xs != X + 1) == (l.jump != 0)));
                    if (!same) {
                        if (getenv("AGX_SIM_STATS")) {
                            const agx_dhit &d = dh[h];
                            fprintf(stderr, "[hostsim] lean mismatch: hit %u tile %u lane %u: decode has %u type %u q %u p0 %u succ %u xs %u | lean has %u last %u q %u p0 %u jump %u | geo %08x qoff %u %u boff %u %u | L %u js %u x %u..%u a_t0 %u b_t0 %u\n",
                                    h, t, lane, a.has, a.type, a.q, a.p0, a.has_succ, a.xs, l.has, l.last, l.q, l.p0, l.jump, lr.geo, lr.qoff1, lr.qoff2, lr.boff1, lr.boff2, d.len, d.jstar, d.x_lo, d.x_hi, d.a_t0, d.b_t0);
                            for (agx_u32 i = 0; i < d.a_nruns; i++) fprintf(stderr, "   a run %u: q %u t %u n %u\n", i, P.runs[d.a_runs + i].q, P.runs[d.a_runs + i].t, P.runs[d.a_runs + i].n);
                            for (agx_u32 i = 0; i < d.b_nruns; i++) fprintf(stderr, "   b run %u: q %u t %u n %u\n", i, P.runs[d.b_runs + i].q, P.runs[d.b_runs + i].t, P.runs[d.b_runs + i].n);
                        }
                        throw Error{E_ARG, "a lean tile record decodes differently from the hit's record"};
                    }
                }
            }
        }
    }
    tile_off[n_tiles] = (agx_u32)tile_hits.size();
    if (getenv("AGX_SIM_STATS")) fprintf(stderr, "[hostsim] lean records: %zu general, %zu one piece, %zu one piece with a jump or without mate positions, %zu two pieces\n", lean_kinds[0], lean_kinds[1], lean_kinds[2], lean_kinds[3]);
    if (getenv("AGX_SIM_STATS")) { size_t lin = 0, cx = 0, cx_hits = 0; for (const agx_dhit &r : tile_recs) (r.a_nruns | r.b_nruns) ? cx++ : lin++; for (const agx_dhit &r : dh) if (!(r.flags & AGX_HF_SKIP) && (r.a_nruns | r.b_nruns)) cx_hits++;
        fprintf(stderr, "[hostsim] tile-list entries: %zu linear pieces, %zu general records (%.1f %%); hits with a multi-run mate: %zu of %zu\n", lin, cx, 100.0 * cx / (lin + cx + 1e-9), cx_hits, dh.size()); }

    // The engine uploads the conti-mer tables as runs (agx_cmseg) and expands them on the device: count = highest rank + 1 per position,
    // scan, fill (agx_k_seg_count / agx_k_seg_fill).  The same element functions here; the result must be the loader's tables, and the hop
    // entries derived from the runs must be the loader's too.
    std::vector<agx_cmkey> cmk(T.cm.size());
    {
        std::vector<agx_u32> cnt((size_t)n_pos + 1, 0), start((size_t)n_pos + 1, 0);
        const agx_u32 n_el = (agx_u32)T.cm.size(), ns = (agx_u32)T.segs.size();
        for (agx_u32 e = 0; e < n_el; e++) { const agx_cmseg &g = T.segs[agx_seg_of_elem(T.segs.data(), ns, e)]; const agx_u32 x = g.pos0 + (e - g.elem0); if (x >= n_pos) throw Error{E_ARG, "conti-mer run beyond the positions"}; cnt[x] = std::max(cnt[x], g.rank + 1); }
        agx_u32 acc = 0; for (agx_u32 x = 0; x <= n_pos; x++) { start[x] = acc; acc += cnt[x]; }
        if (start != T.cm_start) throw Error{E_ARG, "conti-mer runs: cm_start differs"};
        for (agx_u32 e = 0; e < n_el; e++) { const agx_cmseg &g = T.segs[agx_seg_of_elem(T.segs.data(), ns, e)]; const agx_u32 j = e - g.elem0; cmk[start[g.pos0 + j] + g.rank] = agx_cmkey{g.cid, g.coff0 + j * g.dcoff}; }
        for (size_t i = 0; i < T.cm.size(); i++) if (cmk[i].cid != T.cm[i].cid || cmk[i].coff != T.cm[i].coff) throw Error{E_ARG, "conti-mer runs: keys differ"};
        for (agx_u32 x = 0; x < n_pos; x++) {
            const agx_hop h = agx_seg_hop<agx_hop>(T.segs.data(), T.n_seg0, T.cm_start.data(), x);
            if (h.len != T.hop[x].len || (h.len && (h.str_off != T.hop[x].str_off || h.end_pos != T.hop[x].end_pos))) throw Error{E_ARG, "conti-mer runs: hop entries differ"};
        }
    }

    S.node_start.assign(n_pos, 0); S.node_cnt.assign(n_pos, 0); S.pos_succ.assign(n_pos, 0); S.side_pk.assign((size_t)n_pos + 1, 0); S.tile_side.assign((size_t)n_tiles + 1, 0);
    S.reserve((size_t)n_pos * 2 + 1024);
    agx_sweep_args A; memset(&A, 0, sizeof A);
    std::vector<agx_u8> vcodes(P.bases.size());             // the engine stages the read bases as packed classes and expands them on the device at upload
    for (size_t i = 0; i + 3 < P.bases.size(); i += 4) {      // (2-bit classes, then the bases that are not A, C, G, T: agx_pack_classes2)
        const agx_u8 pk = agx_pack_classes2((agx_u8)P.bases[i], (agx_u8)P.bases[i + 1], (agx_u8)P.bases[i + 2], (agx_u8)P.bases[i + 3]);
        for (int b = 0; b < 4; b++) vcodes[i + b] = agx_class_vote_code((pk >> (2 * b)) & 3u);
    }
    for (size_t i = 0; i < P.bases.size(); i++) if (agx_base_class((agx_u8)P.bases[i]) == 4u) vcodes[i] = agx_class_vote_code(4u);
    std::vector<agx_cmhead> cmh((size_t)n_pos + 1);
    for (agx_u32 x = 0; x <= n_pos; x++) agx_cm_head_pos(T.cm_start.data(), cmk.data(), cmh.data(), x, n_pos);
    {   // r03: the device builds the same three tables from the count runs and the conti-mer runs in chunks (agx_k_cm_layout, agx_k_cm_fill): the element
        // functions of those kernels over the host's chunk tables must give exactly the tables above
        CmLayout L; build_cm_layout(T.cm_cnt.data(), n_pos, T.segs.data(), T.segs.size(), L);
        std::vector<agx_u32> start2((size_t)n_pos + 1, 0xDEADBEEFu); std::vector<agx_cmkey> cm2(T.cm.size(), agx_cmkey{0xDEADBEEFu, 0xDEADBEEFu}); std::vector<agx_cmhead> head2((size_t)n_pos + 1, agx_cmhead{1u, 2u, 3u, 4u});
        for (const agx_chunk &c : L.cnt_chunks) { const agx_cntrun &r = L.cnt_runs[c.run]; const agx_u32 n = std::min<agx_u32>(r.len - c.off, AGX_CM_CHUNK); for (agx_u32 j = 0; j < n; j++) agx_cm_layout_pos(r, c.off + j, start2.data(), head2.data()); }
        start2[n_pos] = (agx_u32)T.cm.size(); head2[n_pos] = agx_cmhead{AGX_NONE, AGX_NONE, 0u, 0u};
        for (const agx_chunk &c : L.seg_chunks) { const agx_cmseg &g = T.segs[c.run]; const agx_u32 n = std::min<agx_u32>(g.len - c.off, AGX_CM_CHUNK); for (agx_u32 j = 0; j < n; j++) agx_cm_fill_elem(g, c.off + j, start2.data(), cm2.data(), head2.data()); }
        if (start2 != T.cm_start) throw Error{E_ARG, "chunked conti-mer tables: cm_start differs"};
        if (cm2.size() && memcmp(cm2.data(), cmk.data(), cm2.size() * sizeof(agx_cmkey)) != 0) throw Error{E_ARG, "chunked conti-mer tables: keys differ"};
        if (memcmp(head2.data(), cmh.data(), head2.size() * sizeof(agx_cmhead)) != 0) throw Error{E_ARG, "chunked conti-mer tables: heads differ"};
    }
    A.cm_start = T.cm_start.data(); A.cm = cmk.data(); A.cm_head = cmh.data(); A.ref = T.ref.data();
    A.dhit = dh.data(); A.runs = P.runs.data(); A.vcodes = vcodes.data(); A.stride = P.stride;
    A.tile_off = tile_off.data();
    A.n_pos = n_pos; A.n_tiles = n_tiles; A.k = k; A.iv = iv; A.coverage = coverage;
    auto bind = [&]() {
        A.node_start = S.node_start.data(); A.node_cnt = S.node_cnt.data(); A.pos_succ = S.pos_succ.data(); A.side_pk = S.side_pk.data(); A.tile_side = S.tile_side.data();
        A.nk_cid = S.cid.data(); A.nk_coff = S.coff.data(); A.nk_cid0 = S.cid0.data(); A.nk_coff0 = S.coff0.data(); A.nk_off0 = S.off0.data();
        A.n_base = S.base.data(); A.n_flags = S.flags.data(); A.n_sref = S.sref.data(); A.n_next = S.next.data();
        A.n_counts = S.counts.data(); A.pool_cap = (agx_u32)S.cid.size();
    };
    bind();
    n_big_tiles = 0;
    agx_u32 pool = 0;
    auto get = [&](agx_u32 i) { return tile_recs[i]; };
    std::vector<agx_u32> lds((size_t)AGX_NF * maxv_first * AGX_TILE), big;
    for (agx_u32 t = 0; t < n_tiles; t++) {
        agx_u32 cnt[AGX_TILE], pflag[AGX_TILE]; bool ok = true;
        // what every lane hands to its left neighbour per hit (the kernel does it with a wave shuffle inside the sweep's loop)
        struct Touch { agx_u32 vm, sp; };
        std::vector<Touch> touched[AGX_TILE];
        agx_bucket b{nullptr, AGX_TILE, maxv_first, 0u};
        for (agx_u32 lane = 0; lane < AGX_TILE; lane++) {
            b.base = lds.data() + lane;
            ok &= agx_node_sweep_lane<false>(A, t, t * AGX_TILE + lane, b, cnt[lane], pflag[lane], get, [&](agx_u32 vm, agx_u32 sp) { touched[lane].push_back(Touch{vm, sp}); });
        }
        agx_u32 *store = lds.data(); agx_u32 maxv = maxv_first;
        if (!ok) {                                     // the fallback the engine runs for overflowed tiles
            n_big_tiles++;
            big.assign((size_t)AGX_NF * AGX_MAXV_HUGE * AGX_TILE, 0); store = big.data(); maxv = AGX_MAXV_HUGE;      // (the engine: 4, then 64, then 1024 variants)
            agx_bucket bb{nullptr, AGX_TILE, maxv, 0u};
            for (agx_u32 lane = 0; lane < AGX_TILE; lane++) {
                bb.base = store + lane;
                if (!agx_node_sweep_lane<false>(A, t, t * AGX_TILE + lane, bb, cnt[lane], pflag[lane], get, [](agx_u32, agx_u32) {})) throw Error{E_OVERFLOW, "more than 1024 node variants at one position"};
            }
        }
        agx_u32 total = 0; for (agx_u32 lane = 0; lane < AGX_TILE; lane++) total += cnt[lane];
        if (pool + total > S.cid.size()) { S.reserve((pool + total) * 2); bind(); }
        agx_bucket wb{nullptr, AGX_TILE, maxv, 0u}, wn{nullptr, AGX_TILE, maxv, 0u};
        agx_u32 side_before = 0;                       // the kernel's wave scan
        for (agx_u32 lane = 0; lane < AGX_TILE; lane++) {
            const agx_u32 X = t * AGX_TILE + lane;
            wb.base = store + lane; wn.base = store + lane + 1;
            const agx_u32 ncnt = lane + 1 < AGX_TILE ? cnt[lane + 1] : 0;
            const bool edges = ok && lane < AGX_TILE - 1 && X + 1 < n_pos && cnt[lane] <= AGX_EM_W && ncnt <= AGX_EM_W;
            agx_u32 emask = 0;
            if (edges) for (size_t i = 0; i < touched[lane].size(); i++) agx_edge_merge(emask, touched[lane][i].sp, touched[lane + 1][i].vm);
            const agx_u32 side = agx_node_write_lane(A, X, wb, cnt[lane], pool, pflag[lane], edges, emask, wn, pool + cnt[lane], ncnt);
            if (X < n_pos) S.side_pk[X] = agx_side_pack(side_before, side);
            side_before += side;
            pool += cnt[lane];
        }
        S.tile_side[t] = side_before;
    }
    S.n_nodes = pool;
    // edge build: pass A (lanes = positions) writes the x -> x+1 edges of single-variant positions and collects the slow positions,
    // pass J (lanes = hits) adds the steps that do not go to x+1, pass B resolves every hit of the slow positions' tiles
    auto ins = [&](agx_u32 src, agx_u32 dst) {
        agx_u32 *slots = S.next.data() + (size_t)src * AGX_MAXE;
        for (agx_u32 e = 0; e < AGX_MAXE; e++) { if (slots[e] == AGX_NONE) slots[e] = dst; if (slots[e] == dst) return; }
        S.flags[src] |= AGX_NF_EOVF; S.ovf.push_back(agx_edge_ovf{src, dst});
    };
    std::vector<agx_u32> slow;
    for (agx_u32 X = 0; X < n_pos; X++) {
        const agx_u32 nbs = X + 1 < n_pos ? S.node_start[X + 1] : 0, nbc = X + 1 < n_pos ? S.node_cnt[X + 1] : 0;
        if (agx_edge_fast_lane(A, X, S.node_start[X], S.node_cnt[X], nbs, nbc)) slow.push_back(X);
    }
    for (size_t h = 0; h < dh.size(); h++) agx_edge_jump_hit(A, dh[h], ins);
    for (agx_u32 X : slow) {
        const agx_u32 t = X / AGX_TILE;
        agx_slow_ctx c; agx_edge_slow_ctx(A, X, c);
        agx_u32 pairs = 0;
        for (agx_u32 i = tile_off[t]; i < tile_off[t + 1]; i++) pairs |= agx_edge_slow_pair(A, c, X, tile_recs[i], true, ins);
        for (agx_u32 b = 0; b < AGX_SLOW_V * AGX_SLOW_V; b++) if ((pairs >> b) & 1u) ins(c.s + b / AGX_SLOW_V, c.s1 + b % AGX_SLOW_V);
    }

    // walk preparation, the same per-element functions the compaction kernels run
    agx_compact_args C; memset(&C, 0, sizeof C);
    S.tile_side_start.assign((size_t)n_tiles + 1, 0); S.aid_of.assign((size_t)S.n_nodes + 1, AGX_NONE);      // (side_pk / tile_side were written with the nodes)
    C.node_start = S.node_start.data(); C.node_cnt = S.node_cnt.data(); C.n_flags = S.flags.data(); C.n_base = S.base.data();
    C.nk_off0 = S.off0.data(); C.n_sref = S.sref.data(); C.n_next = S.next.data(); C.ref = T.ref.data(); C.n_pos = n_pos;
    C.side_pk = S.side_pk.data(); C.tile_side_start = S.tile_side_start.data(); C.aid_of = S.aid_of.data();
    agx_u32 run = 0; for (agx_u32 t = 0; t <= n_tiles; t++) { S.tile_side_start[t] = run; run += S.tile_side[t]; }
    S.n_ids = n_pos + run;
    const size_t na = (size_t)S.n_ids + 1;
    S.a_str.assign(na, 0); S.a_meta.assign(na + 64, 0);
    S.a_nid.assign(na, AGX_NONE);
    S.a_ovf.assign(S.ovf.size() + 1, agx_edge_ovf{AGX_NONE, AGX_NONE});
    C.a_str = &S.a_str[0]; C.a_meta = S.a_meta.data(); C.a_nid = S.a_nid.data();
    C.ovf = S.ovf.data(); C.n_ovf = (agx_u32)S.ovf.size(); C.a_ovf = S.a_ovf.data();
    S.a_mark.assign(na + 1, 0); S.side_xpos.assign((size_t)run + 1, 0);
    C.a_mark = S.a_mark.data(); C.side_xpos = S.side_xpos.data(); C.n_ids = S.n_ids;
    C.sparse_min = getenv("AGX_SIM_SPARSE_MIN") ? 1u : 0u;
    for (agx_u32 p : T.chain_end_pos) S.a_mark[p] = 1;                               // (the first threads of agx_k_assign_aid)
    for (agx_u32 x = 0; x < n_pos; x++) agx_assign_aid_pos(C, x);
    for (agx_u32 X = 0; X < n_pos; X++) agx_emit_alive_pos(C, X);
    for (agx_u32 i = 0; i < C.n_ovf; i++) agx_emit_alive_ovf(C, i);
    // sparse record table: bitmap words, rank scan, gather (agx_k_special_bits / scan / agx_k_special_emit)
    const agx_u32 n_words = S.n_ids / 64 + 1;
    S.sp_bits.assign((size_t)n_words + 1, 0); S.sp_cnt.assign((size_t)n_words + 1, 0); S.sp_rank.assign((size_t)n_words + 1, 0);
    for (agx_u32 w = 0; w < n_words; w++) {
        unsigned long long bits = 0;
        for (agx_u32 l = 0; l < 64; l++) if (agx_special_id(C, w * 64 + l)) bits |= 1ull << l;
        S.sp_bits[w] = bits; S.sp_cnt[w] = (agx_u32)__builtin_popcountll(bits);
    }
    agx_u32 acc = 0; for (agx_u32 w = 0; w <= n_words; w++) { S.sp_rank[w] = acc; acc += S.sp_cnt[w]; }
    S.n_special = S.sp_rank[n_words]; S.sp_node.resize((size_t)S.n_special + 1); S.sp_hop.resize((size_t)S.n_special + 1);
    for (agx_u32 w = 0; w < n_words; w++) for (agx_u32 l = 0; l < 64; l++) if ((S.sp_bits[w] >> l) & 1ull)
    {
        const agx_u32 a = w * 64 + l, at = S.sp_rank[w] + (agx_u32)__builtin_popcountll(S.sp_bits[w] & ((1ull << l) - 1ull));
        S.sp_node[at] = agx_walk_record(C, a);
        S.sp_hop[at] = T.hop[a < n_pos ? a : S.side_xpos[a - n_pos]];
    }
    S.walk_args = C;
}

char *dup_buf(const std::string &s) { char *p = (char *)malloc(s.size() + 1); memcpy(p, s.data(), s.size()); p[s.size()] = 0; return p; }

}  // namespace

extern "C" {

typedef struct {
    char *initial_contigs; size_t initial_len;
    char *pre_extended; size_t pre_len;
    char *extended; size_t extended_len;
    char error[256];
    uint32_t n_pos, n_nodes, n_edges; int32_t n_big_tiles;
    uint32_t *node_start;   // canonical (position-ordered) numbering, same layout as the oracle's dump
    uint32_t *node_key; int32_t *node_cnt; uint32_t *node_slen; uint32_t *edge_start; uint32_t *edge_dst;   // edge_dst sorted per node
    uint64_t n_walk_ids, n_special, n_fetched;   // walk graph: ids, records in the sparse table, records read through the fetch hook
} agx_hostsim_result;

int agx_hostsim_run_unit(const char *tmp_dir, int unit, int k, int iv, int coverage, long batch, int maxv_first, int want_graph, agx_hostsim_result *out) {
    memset(out, 0, sizeof *out);
    try {
        const std::string d = tmp_dir, u = std::to_string(unit);
        Threads T; Pairs P;
        load_unit_reference(d + "/_genome." + u + ".fa", T.ref);
        thread_contigs_from_files(d + "/_contigs.fa", d + "/_contigs_genome." + u + ".psl", T);
        if (getenv("AGX_SIM_READS_INDEX")) {            // the loader path of agx_unit_load_files_shared
            ReadsIndex *ri = reads_index_open(d + "/_reads.fa");
            try { load_pairs_from_files(d + "/_reads.fa", d + "/_reads_genome." + u + ".bowtie", batch, (agx_u32)k, P, ri); } catch (...) { reads_index_close(ri); throw; }
            reads_index_close(ri);
        } else load_pairs_from_files(d + "/_reads.fa", d + "/_reads_genome." + u + ".bowtie", batch, (agx_u32)k, P);
        SimGraph S; int nbig = 0;
        simulate(T, P, (agx_u32)k, iv, coverage, maxv_first > 0 ? (agx_u32)maxv_first : AGX_MAXV_LDS, S, nbig);
        GraphView G; G.n_pos = (agx_u32)T.ref.size(); G.n_ids = S.n_ids;
        G.meta = S.a_meta.data(); G.str = S.a_str.data(); G.side_xpos = S.side_xpos.data();
        G.sp_bits = S.sp_bits.data(); G.sp_rank = S.sp_rank.data(); G.sp_node = S.sp_node.data(); G.sp_hop = S.sp_hop.data(); G.n_special = S.n_special;
        G.fetch = [](void *ctx, agx_u32 first, agx_u32 stride, agx_u32 rows, agx_u32 width, agx_walknode *o) {
            for (agx_u32 r = 0; r < rows; r++) for (agx_u32 c = 0; c < width; c++) { const SimGraph *g = (const SimGraph *)ctx; const size_t a = (size_t)first + (size_t)r * stride + c; if (a >= g->n_ids) throw Error{E_ARG, "record fetch beyond the walk graph"}; o[(size_t)r * width + c] = agx_walk_record(g->walk_args, (agx_u32)a); }
        }; G.fetch_ctx = &S;
        G.ovf = S.a_ovf.data(); G.n_ovf = S.ovf.size();
        if (const char *rep = getenv("AGX_WALK_REPEAT")) {           // walk micro-benchmark: best of N on an already built graph
            double best = 1e30;
            for (int i = 0; i < atoi(rep); i++) { UnitOutput Q; const auto t0 = std::chrono::steady_clock::now(); walk_join_scaffold(view_of(T, P), G, Q); const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); if (ms < best) best = ms; }
            fprintf(stderr, "[hostsim] walk+join+scaffold best of %s: %.2f ms\n", rep, best);
        }
        // AGX_SIM_STREAM=<pieces>[,<microseconds per piece>]: the walk graph arrives the way the engine's streamed download delivers it (GraphView::wait_landed) — the arrays the walk
        // is given are full of junk and a thread copies the real ones in, a position window at a time from the front (main ids and the side ids of their positions; the bases last).
        // A walker that looks at anything it has not waited for reads junk, and the outputs differ from the oracle's.
        struct Landing {
            const SimGraph &S; agx_u32 n_pos, n_ids; int pieces; long us;
            std::vector<agx_u8> meta; std::vector<char> str; std::vector<unsigned long long> bits; std::vector<agx_u32> rank; std::vector<agx_walknode> node; std::vector<agx_hop> hop;
            std::vector<agx_u32> cut_main, cut_side;      // frontier after piece w: main ids below cut_main[w + 1], side ids below cut_side[w + 1]
            std::atomic<int> landed{0}; std::atomic<bool> str_in{false}; std::thread th;
            Landing(const SimGraph &g, agx_u32 np, int n, long u) : S(g), n_pos(np), n_ids(g.n_ids), pieces(n), us(u) {
                meta.assign(S.a_meta.size(), 0x5A); str.assign(S.a_str.size(), '?'); bits.assign(S.sp_bits.size(), 0xDEADBEEFDEADBEEFull); rank.assign(S.sp_rank.size(), 0x7FFFFFF0u);
                agx_walknode junk; memset(&junk, 0xEE, sizeof junk); node.assign(S.sp_node.size(), junk); agx_hop hj; memset(&hj, 0xEE, sizeof hj); hop.assign(S.sp_hop.size(), hj);
                const agx_u32 n_side = n_ids - n_pos;
                for (int w = 0; w <= pieces; w++) {
                    const agx_u32 x = w == pieces ? n_pos : (agx_u32)((unsigned long long)n_pos * (unsigned)w / (unsigned)pieces) & ~63u;
                    cut_main.push_back(x); cut_side.push_back(w == pieces ? n_ids : n_pos + (agx_u32)(std::lower_bound(S.side_xpos.begin(), S.side_xpos.begin() + n_side, x) - S.side_xpos.begin()));
                }
            }
            agx_u32 rank_of(agx_u32 a) const { return a >= n_ids ? S.n_special : S.sp_rank[a >> 6] + (agx_u32)__builtin_popcountll(S.sp_bits[a >> 6] & ((1ull << (a & 63u)) - 1ull)); }
            void ids(agx_u32 lo, agx_u32 hi) {          // everything about walk ids [lo, hi)
                if (lo >= hi) return;
                memcpy(meta.data() + lo, S.a_meta.data() + lo, hi - lo);
                for (agx_u32 w = lo >> 6; w <= (hi - 1) >> 6; w++) { bits[w] = S.sp_bits[w]; rank[w] = S.sp_rank[w]; }
                for (agx_u32 r = rank_of(lo), re = rank_of(hi); r < re; r++) { node[r] = S.sp_node[r]; hop[r] = S.sp_hop[r]; }
            }
            void start() {
                th = std::thread([this] {
                    for (int w = 0; w < pieces; w++) {
                        if (us) std::this_thread::sleep_for(std::chrono::microseconds(us));
                        ids(cut_main[(size_t)w], cut_main[(size_t)w + 1]); ids(cut_side[(size_t)w], cut_side[(size_t)w + 1]);
                        if (w + 1 == pieces) memcpy(meta.data() + n_ids, S.a_meta.data() + n_ids, meta.size() - n_ids);      // (the padding behind the table)
                        landed.store(w + 1, std::memory_order_release);
                    }
                    if (us) std::this_thread::sleep_for(std::chrono::microseconds(us));
                    memcpy(str.data(), S.a_str.data(), str.size()); str_in.store(true, std::memory_order_release);
                });
            }
            static void wait_landed(void *ctx, agx_u32 main_hi, agx_u32 side_hi) {
                Landing *L = (Landing *)ctx; int need = 0;
                while (need < L->pieces && (L->cut_main[(size_t)need] < main_hi || L->cut_side[(size_t)need] < side_hi)) need++;
                while (L->landed.load(std::memory_order_acquire) < need) std::this_thread::yield();
            }
            static void wait_str(void *ctx) { Landing *L = (Landing *)ctx; while (!L->str_in.load(std::memory_order_acquire)) std::this_thread::yield(); }
            ~Landing() { if (th.joinable()) th.join(); }
        };
        std::unique_ptr<Landing> landing;
        if (const char *e = getenv("AGX_SIM_STREAM")) {
            const int n = std::max(1, atoi(e)); const char *c = strchr(e, ',');
            landing.reset(new Landing(S, G.n_pos, n, c ? atol(c + 1) : 0));
            G.meta = landing->meta.data(); G.str = landing->str.data(); G.sp_bits = landing->bits.data(); G.sp_rank = landing->rank.data(); G.sp_node = landing->node.data(); G.sp_hop = landing->hop.data();
            G.meta_rw = landing->meta.data();            // (the engine's download buffer is the first walker's own array)
            G.wait_landed = Landing::wait_landed; G.wait_str = Landing::wait_str; G.land_ctx = landing.get(); G.land_ms = std::max(0.001, 1e-3 * (double)landing->us * n);
            landing->start();
        }
        // AGX_SIM_ASSISTANT=1: with a second thread for the outputs, as the engine runs it (the written records are formatted while the walk goes on)
        struct ThreadAssistant : Assistant {
            std::thread t[15];
            void run(std::function<void()> f, int who) override { wait(who); t[who] = std::thread(std::move(f)); }
            void wait(int who) override { if (t[who].joinable()) t[who].join(); }
            int helpers() const override { return 15; }
            ~ThreadAssistant() override { for (int i = 0; i < 15; i++) wait(i); }
        } second;
        UnitOutput O; walk_join_scaffold(view_of(T, P), G, O, getenv("AGX_SIM_ASSISTANT") ? &second : nullptr);
        out->initial_contigs = dup_buf(T.initial_contigs); out->initial_len = T.initial_contigs.size();
        out->pre_len = O.pre_extended.n; out->pre_extended = O.pre_extended.release();
        out->extended_len = O.extended.n; out->extended = O.extended.release();
        out->n_big_tiles = nbig; out->n_walk_ids = S.n_ids; out->n_special = S.n_special; out->n_fetched = O.n_fetched;
        if (want_graph) {
            const agx_u32 n_pos = G.n_pos, nn = S.n_nodes;
            out->n_pos = n_pos; out->n_nodes = nn;
            out->node_start = (uint32_t *)malloc(4 * ((size_t)n_pos + 1));
            std::vector<agx_u32> canon(nn);
            agx_u32 id = 0;
            for (agx_u32 x = 0; x < n_pos; x++) { out->node_start[x] = id; for (agx_u32 v = 0; v < S.node_cnt[x]; v++) canon[S.node_start[x] + v] = id++; }
            out->node_start[n_pos] = id;
            out->node_key = (uint32_t *)malloc(4 * 6 * ((size_t)nn + 1)); out->node_cnt = (int32_t *)malloc(4 * 6 * ((size_t)nn + 1));
            out->node_slen = (uint32_t *)malloc(4 * ((size_t)nn + 1)); out->edge_start = (uint32_t *)malloc(4 * ((size_t)nn + 1));
            std::vector<std::vector<agx_u32> > adj(nn);
            for (agx_u32 v = 0; v < nn; v++) for (agx_u32 e = 0; e < AGX_MAXE; e++) if (S.next[(size_t)v * AGX_MAXE + e] != AGX_NONE) adj[canon[v]].push_back(canon[S.next[(size_t)v * AGX_MAXE + e]]);
            for (const agx_edge_ovf &e : S.ovf) adj[canon[e.src]].push_back(canon[e.dst]);
            size_t ne = 0;
            for (auto &a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); ne += a.size(); }
            out->n_edges = (uint32_t)ne; out->edge_dst = (uint32_t *)malloc(4 * (ne + 1));
            size_t eo = 0;
            for (agx_u32 v = 0; v < nn; v++) {
                const agx_u32 c = canon[v];
                uint32_t *kk = out->node_key + 6 * (size_t)c;
                kk[0] = S.cid[v]; kk[1] = S.coff[v]; kk[2] = S.cid0[v]; kk[3] = S.coff0[v]; kk[4] = S.off0[v] == AGX_NONE ? AGX_NONE : 0; kk[5] = S.off0[v];
                memcpy(out->node_cnt + 6 * (size_t)c, S.counts.data() + 6 * (size_t)v, 24);
                out->node_slen[c] = (S.sref[v].qlen >> 16) & 0x7FFF;
            }
            for (agx_u32 c = 0; c < nn; c++) { out->edge_start[c] = (uint32_t)eo; for (agx_u32 d2 : adj[c]) out->edge_dst[eo++] = d2; }
            out->edge_start[nn] = (uint32_t)eo;
        }
        return 0;
    } catch (const Error &e) {
        snprintf(out->error, sizeof out->error, "%s", e.msg.c_str());
        return e.code ? e.code : -1;
    }
}

void agx_hostsim_free(agx_hostsim_result *r) {
    free(r->initial_contigs); free(r->pre_extended); free(r->extended);
    free(r->node_start); free(r->node_key); free(r->node_cnt); free(r->node_slen); free(r->edge_start); free(r->edge_dst);
    memset(r, 0, sizeof *r);
}

}  // extern "C"

// ---- fast loaders against the general loaders (tests/test_fast_loader.py) -----------------------------------------------------------------
// Loads one unit's text files both ways and compares everything the engine takes from either.  Returns a bit mask of which fast loader declined
// (1: contig threading, 2: read alignments; both are then served by the general path in the product), or a negative number with a message when
// the two disagree — including "the general loader reports an error and the fast one accepts the input".
namespace {
struct VecSink : StageSink { std::vector<char> buf[SA_N]; void *take(int w, size_t b) override { buf[w].assign(b + 64, 0); return buf[w].data(); } };
}
extern "C" int agx_hostsim_compare_loaders(const char *tmp_dir, int unit, int k, long batch, int threads, char *msg, size_t msg_len) {
    auto say = [&](const std::string &m) { if (msg && msg_len) snprintf(msg, msg_len, "%s", m.c_str()); };
    say("");
    const std::string d = tmp_dir, s = std::to_string(unit);
    int declined = 0;
    try {
        // ---- contigs ----
        Threads T1, T2; bool slow_ok = true; std::string slow_err;
        std::string ref;
        try { load_unit_reference(d + "/_genome." + s + ".fa", ref); } catch (const Error &e) { say("genome: " + e.msg); return 64; }
        T1.ref = ref; T2.ref = ref;
        try { thread_contigs_from_files(d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", T1); } catch (const Error &e) { slow_ok = false; slow_err = e.msg; }
        bool fast_ok = false;
        try { fast_ok = thread_contigs_fast(d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", T2); } catch (const Error &e) { if (slow_ok) { say("fast contig threading threw: " + e.msg); return -1; } }
        if (!slow_ok && fast_ok) { say("general contig threading fails (" + slow_err + ") but the fast one accepts the input"); return -2; }
        if (!fast_ok) { declined |= 1; if (T2.ref != ref) { say("fast contig threading declined but changed the reference"); return -3; } }
        if (slow_ok && fast_ok) {
            auto bad = [&](const char *what) { say(std::string("contig threading differs: ") + what); return -10; };
            if (T1.ref != T2.ref) return bad("ref");
            if (T1.n_ref != T2.n_ref) return bad("n_ref");
            if (T1.n_cm != T2.n_cm) return bad("n_cm");
            if (T1.cm_cnt != T2.cm_cnt) return bad("cm_cnt");
            if (T1.n_seg0 != T2.n_seg0) return bad("n_seg0");
            if (T1.segs.size() != T2.segs.size()) return bad("number of runs");
            for (size_t i = 0; i < T1.segs.size(); i++) if (memcmp(&T1.segs[i], &T2.segs[i], sizeof(agx_cmseg)) != 0) { char b[256]; const agx_cmseg &x = T1.segs[i], &y = T2.segs[i];
                snprintf(b, sizeof b, "run %zu: general (pos0 %u len %u cid %u coff0 %u dcoff %u rank %u str0 %u len0 %u end %u elem0 %u) fast (%u %u %u %u %u %u %u %u %u %u)", i,
                         x.pos0, x.len, x.cid, x.coff0, x.dcoff, x.rank, x.hop_str0, x.hop_len0, x.hop_end, x.elem0, y.pos0, y.len, y.cid, y.coff0, y.dcoff, y.rank, y.hop_str0, y.hop_len0, y.hop_end, y.elem0); return bad(b); }
            if (T1.chain_str != T2.chain_str) return bad("chain_str");
            if (T1.chain_off != T2.chain_off) return bad("chain_off");
            if (T1.chain_end_pos != T2.chain_end_pos) return bad("chain_end_pos");
            if (T1.initial_contigs != T2.initial_contigs) return bad("initial_contigs");
        }
        // ---- read alignments ----
        std::unique_ptr<ReadsIndex, void (*)(ReadsIndex *)> ri(nullptr, reads_index_close);
        try { ri.reset(reads_index_open(d + "/_reads.fa")); } catch (const Error &e) { say("reads: " + e.msg); return 64; }
        Pairs P; VecSink A, B; StagedPairs SA, SB; slow_ok = true; std::vector<agx_hit> staged;
        try { load_pairs_from_files(d + "/_reads.fa", d + "/_reads_genome." + s + ".bowtie", batch, (agx_u32)k, P, ri.get()); stage_pairs(P, (agx_u32)k, (unsigned)threads, A, SA, &staged); } catch (const Error &e) { slow_ok = false; slow_err = e.msg; }
        fast_ok = false;
        try { fast_ok = load_pairs_fast(*ri, d + "/_reads_genome." + s + ".bowtie", batch, (agx_u32)k, (unsigned)threads, B, SB); } catch (const Error &e) { if (slow_ok) { say("fast read loader threw: " + e.msg); return -4; } }
        if (!slow_ok && fast_ok) { say("general read loader fails (" + slow_err + ") but the fast one accepts the input"); return -5; }
        if (!fast_ok) declined |= 2;
        if (slow_ok) {      // pass J's list: exactly the hits whose left mate has several runs
            size_t at = 0;
            for (size_t i = 0; i < staged.size(); i++) if (((staged[i].pad[0] & 1u) ? staged[i].nruns2 : staged[i].nruns1) >= 2) { if (at >= SA.n_jump || SA.jump[at] != i) { say("list of hits for pass J is wrong at hit " + std::to_string(i)); return -33; } at++; }
            if (at != SA.n_jump) { say("list of hits for pass J is too long"); return -33; }
        }
        if (slow_ok) {      // the wire forms (agx_core.h) must unpack to exactly the staged hits and the loader's runs
            for (size_t i = 0; i < SA.nh; i++) { const agx_hit h = agx_unpack_hit(SA.hits[i], SA.sides); agx_hit w = staged[i]; w.pad[1] = w.pad[2] = 0; if (memcmp(&h, &w, sizeof h) != 0) { say("wire form of hit " + std::to_string(i) + " does not unpack to the staged hit"); return -30; } }
            for (size_t i = 0; i < SA.n_runs; i++) if (SA.runs[i].q != P.runs[i].q || SA.runs[i].t != P.runs[i].t || SA.runs[i].n != P.runs[i].n) { say("wire form of run " + std::to_string(i) + " differs"); return -31; }
        }
        if (slow_ok && fast_ok) {
            auto bad = [&](const std::string &what) { say("read alignments differ: " + what); return -20; };
            if (SA.nh != SB.nh) return bad("number of hits " + std::to_string(SA.nh) + " / " + std::to_string(SB.nh));
            if (SA.n_runs != SB.n_runs) return bad("number of runs");
            if (SA.n_sides != SB.n_sides) return bad("number of side records");
            if (SA.n_sam_pairs != SB.n_sam_pairs) return bad("n_sam_pairs " + std::to_string(SA.n_sam_pairs) + " / " + std::to_string(SB.n_sam_pairs));
            if (SA.n_pairs_in_file != SB.n_pairs_in_file) return bad("n_pairs_in_file");
            if (SA.nh) { if (SA.stride != SB.stride) return bad("stride"); if (SA.maxlen != SB.maxlen) return bad("maxlen"); }
            if (SA.n_rows != SB.n_rows) return bad("rows " + std::to_string(SA.n_rows) + " / " + std::to_string(SB.n_rows));
            for (size_t i = 0; i < SA.nh; i++) if (memcmp(&SA.hits[i], &SB.hits[i], sizeof(agx_whit)) != 0) { char b[256]; const agx_hit x = agx_unpack_hit(SA.hits[i], SA.sides), y = agx_unpack_hit(SB.hits[i], SB.sides);
                snprintf(b, sizeof b, "hit %zu: general (row %u pos %u %u runs %u+%u %u+%u len %u rev %u%u back %u left %u) fast (row %u pos %u %u runs %u+%u %u+%u len %u rev %u%u back %u left %u)", i,
                         x.slot1, x.pos1, x.pos2, x.runs1, x.nruns1, x.runs2, x.nruns2, x.len, x.rev1, x.rev2, x.back, x.pad[0], y.slot1, y.pos1, y.pos2, y.runs1, y.nruns1, y.runs2, y.nruns2, y.len, y.rev1, y.rev2, y.back, y.pad[0]); return bad(b); }
            if (SA.n_sides && memcmp(SA.sides, SB.sides, SA.n_sides * sizeof(agx_wside)) != 0) return bad("side records");
            if (SA.n_runs && memcmp(SA.runs, SB.runs, SA.n_runs * sizeof(agx_wrun)) != 0) return bad("runs");
            if (SA.n_jump != SB.n_jump || (SA.n_jump && memcmp(SA.jump, SB.jump, SA.n_jump * 4) != 0)) return bad("list of hits for pass J");
            if (SA.n_codes != SB.n_codes || (SA.n_codes && memcmp(SA.codes, SB.codes, SA.n_codes) != 0)) return bad("codes");
            if (SA.n_other != SB.n_other || (SA.n_other && memcmp(SA.other, SB.other, SA.n_other * 8) != 0)) return bad("list of other bases");
            std::vector<agx_u16> row_len(SA.n_rows, 0);
            for (size_t i = 0; i < SA.nh; i++) row_len[SA.hits[i].row] = std::max(row_len[SA.hits[i].row], SA.hits[i].len);
            if (SB.row_off.size() != SA.n_rows || SA.row_slot.size() != SA.n_rows) return bad("row tables");
            for (size_t r = 0; r < SA.n_rows; r++) if (memcmp(P.bases.data() + (size_t)SA.row_slot[r] * P.stride, ri->fv.p + SB.row_off[r], row_len[r]) != 0) return bad("bases of row " + std::to_string(r));
        }
        // the reference bases in their wire form: what the device would expand them to
        {
            std::vector<agx_u8> packed((ref.size() + 3) / 4 + 64); std::vector<agx_refx> others;
            if (pack_reference(ref.data(), ref.size(), (unsigned)threads, packed.data(), others)) {
                std::string back(ref.size(), '?');
                for (size_t x = 0; x < ref.size(); x++) back[x] = agx_ref_base((packed[x >> 2] >> (2 * (x & 3))) & 3u);
                for (const agx_refx &o : others) for (agx_u32 j = 0; j < o.len; j++) back[(size_t)o.pos + j] = (char)o.byte;
                if (back != ref) { say("reference bases do not survive their wire form"); return -32; }
            }
        }
    } catch (const Error &e) { say("unexpected: " + e.msg); return -99; }
    catch (const std::exception &e) { say(std::string("unexpected: ") + e.what()); return -99; }
    return declined;
}

// tmp/_agx_pairs.<u>.bin (a unit's read alignments handed over staged, agx_host.h) against what the general loader + staging make of the unit's TEXT files: every array and
// every count, byte for byte; and the listed other bases must carry the bytes the reads have there.  0 = equal; negative: msg says what differs.
extern "C" int agx_hostsim_compare_staged(const char *tmp_dir, int unit, int k, long batch, int threads, char *msg, size_t msg_len) {
    auto say = [&](const std::string &m) { if (msg && msg_len) snprintf(msg, msg_len, "%s", m.c_str()); };
    try {
        const std::string d = tmp_dir, s = std::to_string(unit);
        struct VSink : StageSink { std::vector<std::vector<char>> keep; void *take(int, size_t bytes) override { keep.emplace_back(bytes + 64); return keep.back().data(); } } A;
        Pairs P; StagedPairs S;
        load_pairs_from_files(d + "/_reads.fa", d + "/_reads_genome." + s + ".bowtie", batch, (agx_u32)k, P, nullptr);
        stage_pairs(P, (agx_u32)k, (unsigned)threads, A, S);
        PairsFile F;
        if (!open_pairs_file(pairsfile::path_of(d, unit), F)) { say("no staged-pairs file"); return -1; }
        using namespace pairsfile;
        const Header &H = F.H;
        auto bad = [&](const std::string &what) { say("staged pairs differ from the text path: " + what); return -20; };
        if (H.k != (agx_u32)k || H.batch != (agx_u32)(batch <= 0 ? 1000000 : batch)) return bad("k / BATCH in the header");
        if (H.nh != S.nh) return bad("number of hits " + std::to_string(H.nh) + " / " + std::to_string(S.nh));
        if (H.n_runs != S.n_runs || H.n_sides != S.n_sides || H.n_jump != S.n_jump) return bad("number of runs / side records / pass-J hits");
        if (H.pairs_in_file != S.n_pairs_in_file || H.sam_pairs != S.n_sam_pairs) return bad("pairs in the reads file / SAM line pairs " + std::to_string(H.sam_pairs) + " / " + std::to_string(S.n_sam_pairs));
        if (S.nh && (H.stride != S.stride || H.maxlen != S.maxlen)) return bad("stride / longest read");
        if (H.n_rows != S.n_rows || H.n_codes != S.n_codes || H.n_other != S.n_other) return bad("rows / codes / other bases");
        if (S.nh && memcmp(F.sec(S_HITS), S.hits, S.nh * sizeof(agx_whit)) != 0) return bad("hits");
        if (S.n_sides && memcmp(F.sec(S_SIDES), S.sides, S.n_sides * sizeof(agx_wside)) != 0) return bad("side records");
        if (S.n_runs && memcmp(F.sec(S_RUNS), S.runs, S.n_runs * sizeof(agx_wrun)) != 0) return bad("runs");
        if (S.n_jump && memcmp(F.sec(S_JUMP), S.jump, S.n_jump * 4) != 0) return bad("list of hits for pass J");
        if (S.n_codes && memcmp(F.sec(S_CODES), S.codes, S.n_codes) != 0) return bad("codes");
        if (S.n_other && memcmp(F.sec(S_OTHER), S.other, S.n_other * 8) != 0) return bad("list of other bases");
        {   // the hits in tile order (order_hits, r05), against what the device's hit preparation will find: a permutation; the tile of every kept hit's first arrival (agx_hit_prep's x_lo)
            // is its key; keys ascend, hit numbers ascend inside a key; tile_first counts; pass J's list names the same hits; every tile's list is what a filter over its
            // window of the order (+ the long hits) gives — the very predicate agx_k_tile_fill applies
            const size_t n_pos = [&] { std::string ref; load_unit_reference(d + "/_genome." + s + ".fa", ref); return ref.size(); }();
            const agx_u32 n_tiles = (agx_u32)((n_pos + AGX_TILE - 1) / AGX_TILE);
            std::vector<agx_u32> perm(S.nh + 1), first((size_t)n_tiles + 2), jump_at(S.n_jump + 1);
            order_hits(S.hits, S.nh, S.sides, S.runs, S.jump, S.n_jump, n_pos, (unsigned)threads, perm.data(), first.data(), jump_at.data());
            std::vector<agx_run> runs(S.n_runs + 1); for (size_t i = 0; i < S.n_runs; i++) runs[i] = agx_run{S.runs[i].q, S.runs[i].t, S.runs[i].n};
            std::vector<agx_u8> seen(S.nh, 0); std::vector<agx_dhit> dh(S.nh);
            const agx_u32 lookback = agx_tile_lookback(S.maxlen, (agx_u32)k);
            agx_u32 prev_t = 0, prev_h = 0; size_t jat = 0;
            std::vector<agx_u32> longs;
            for (size_t i = 0; i < S.nh; i++) {
                const agx_u32 h = perm[i];
                if (h >= S.nh || seen[h]) return bad("hits in tile order: not a permutation");
                seen[h] = 1;
                const agx_hit H = agx_unpack_hit(S.hits[h], S.sides);
                (void)agx_hit_prep(H, false, (H.pad[0] & 1u) != 0, H.slot1, runs.data(), (agx_u32)k, dh[i]);
                agx_u32 tl = agx_whit_first_x(S.hits[h], S.sides, S.runs) / AGX_TILE; if (tl >= n_tiles) tl = n_tiles - 1;
                const bool kept = !(dh[i].flags & AGX_HF_SKIP) && dh[i].x_hi < n_pos;
                if (kept && dh[i].x_lo / AGX_TILE != tl) return bad("hits in tile order: hit " + std::to_string(h) + " is keyed by tile " + std::to_string(tl) + " but its first arrival is at " + std::to_string(dh[i].x_lo));
                if (i && (tl < prev_t || (tl == prev_t && h < prev_h))) return bad("hits in tile order: the order is not (tile, hit number)");
                if (!(first[tl] <= i && i < first[tl + 1])) return bad("hits in tile order: tile_first does not hold hit " + std::to_string(h));
                prev_t = tl; prev_h = h;
                const bool is_jump = ((H.pad[0] & 1u) ? H.nruns2 : H.nruns1) >= 2;
                if (is_jump) { if (jat >= S.n_jump || jump_at[jat] != i) return bad("hits in tile order: pass J's list"); jat++; }
                if (kept && dh[i].x_hi / AGX_TILE - dh[i].x_lo / AGX_TILE >= lookback) longs.push_back((agx_u32)i);
            }
            if (jat != S.n_jump || first[0] != 0 || first[n_tiles] != S.nh) return bad("hits in tile order: counts");
            for (agx_u32 t = 0; t < n_tiles; t++) {      // the window filter against the definition (every kept hit whose span holds the tile)
                std::vector<agx_u32> want, got;
                for (size_t i = first[t >= lookback - 1 ? t - (lookback - 1) : 0]; i < first[t + 1]; i++) {
                    const bool kept = !(dh[i].flags & AGX_HF_SKIP) && dh[i].x_hi < n_pos, lng = kept && dh[i].x_hi / AGX_TILE - dh[i].x_lo / AGX_TILE >= lookback;
                    if (kept && !lng && dh[i].x_hi / AGX_TILE >= t) got.push_back(perm[i]);
                }
                for (agx_u32 i : longs) if (dh[i].x_lo / AGX_TILE <= t && t <= dh[i].x_hi / AGX_TILE) got.push_back(perm[i]);
                std::sort(got.begin(), got.end());
                for (size_t i = 0; i < S.nh; i++) if (!(dh[i].flags & AGX_HF_SKIP) && dh[i].x_hi < n_pos && dh[i].x_lo / AGX_TILE <= t && t <= dh[i].x_hi / AGX_TILE) want.push_back(perm[i]);
                std::sort(want.begin(), want.end());
                if (want != got) return bad("hits in tile order: the window of tile " + std::to_string(t) + " does not give its list");
                if (n_tiles > 2000 && t > 600 && t + 600 < n_tiles) t += 37;      // (large units: a sample of the tiles — the check is quadratic)
            }
        }
        const std::vector<agx_u8> ob = other_bytes_of(P, S);
        if (S.n_other && memcmp(F.sec(S_OTHERB), ob.data(), S.n_other) != 0) return bad("bytes of the other bases");
        // every base of every row, decoded the way the walk decodes a k-mer tail from the 2-bit rows, is the read's base
        const agx_u8 *codes = (const agx_u8 *)F.sec(S_CODES); const unsigned long long *oi = (const unsigned long long *)F.sec(S_OTHER); const agx_u8 *obf = (const agx_u8 *)F.sec(S_OTHERB);
        std::vector<agx_u16> row_len(S.n_rows, 0);
        for (size_t i = 0; i < S.nh; i++) row_len[S.hits[i].row] = std::max(row_len[S.hits[i].row], S.hits[i].len);
        for (size_t r = 0; r < S.n_rows; r++) for (agx_u32 j = 0; j < row_len[r]; j++) {
            char c = "ACGT"[(codes[r * (H.stride / 4) + (j >> 2)] >> (2u * (j & 3u))) & 3u];
            if (c == 'A' && H.n_other) { const unsigned long long key = (unsigned long long)r * H.stride + j, *e = oi + H.n_other, *at = std::lower_bound(oi, e, key); if (at != e && *at == key) c = (char)obf[at - oi]; }
            if (c != P.bases[(size_t)S.row_slot[r] * P.stride + j]) return bad("base " + std::to_string(j) + " of row " + std::to_string(r) + " does not come back out of the 2-bit rows");
        }
        return 0;
    } catch (const Error &e) { say("unexpected: " + e.msg); return -99; }
    catch (const std::exception &e) { say(std::string("unexpected: ") + e.what()); return -99; }
}

// ---- the CPU quota reader (tests/test_host_misc.py): a made-up /proc/self/cgroup and /sys/fs/cgroup tree -------------------------------------
extern "C" unsigned agx_hostsim_cgroup_quota(const char *proc_cgroup, const char *sys_root) { return agx::cgroup_cpu_quota(proc_cgroup, sys_root); }
extern "C" unsigned agx_hostsim_usable_cpus() { return agx::usable_cpus(); }

// ---- the read rows' upload form (tests/test_row_diffs.py): build_row_diffs against the decoder the device runs (agx_row_chunk16), on made-up rows ------
// Rows are cut from a random reference under a random anchor geometry (either strand, either mate as the left one, simple or with runs, inside the unit or
// hanging over its end), mutated at `mut_permille`; `dirty_tail` fills the bases beyond a read's length with noise (staging leaves zeros there: the codec must be
// exact anyway).  Every row is decoded the way agx_k_expand_rows does — anchor bit -> hit -> block offset + counts in front of it -> chunks — and compared
// with its 2-bit classes.  Returns 0 and the unit / explicit-row counts, or the first mismatch in msg.
extern "C" int agx_hostsim_rowdiff_roundtrip(unsigned seed, unsigned n_pos, unsigned n_rows, unsigned stride, unsigned maxlen, unsigned mut_permille, int dirty_tail, unsigned threads,
                                             unsigned long long *n_units, unsigned long long *n_explicit, char *msg, size_t msg_len) {
    auto say = [&](const std::string &m) { if (msg && msg_len) snprintf(msg, msg_len, "%s", m.c_str()); };
    say("");
    try {
        uint64_t st = seed * 0x9E3779B97F4A7C15ull + 12345;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (agx_u32)(st >> 11); };
        std::string ref(n_pos, 'A');
        for (auto &c : ref) { const agx_u32 r = rnd() % 1000; c = r < 3 ? 'N' : "ACGT"[r & 3]; }
        std::vector<agx_u8> packed((n_pos + 3) / 4 + 64, 0xA5);
        std::vector<agx_refx> others;
        if (!pack_reference(ref.data(), n_pos, 2, packed.data(), others)) { say("reference did not pack"); return 2; }
        std::vector<agx_u32> wref(packed.size() / 4 + 1); memcpy(wref.data(), packed.data(), packed.size());
        const size_t row_bytes = stride / 4;
        std::vector<agx_u8> codes((size_t)n_rows * row_bytes + 16, 0);
        std::vector<agx_whit> hits; std::vector<agx_wside> sides; std::vector<agx_wrun> runs;
        std::vector<agx_u32> order(n_rows);
        for (agx_u32 r = 0; r < n_rows; r++) order[r] = r;
        if ((seed & 3u) == 1u) for (agx_u32 r = n_rows; r > 1; r--) std::swap(order[r - 1], order[rnd() % r]);      // rows not in hit order: the form must be declined
        for (agx_u32 i = 0; i < n_rows; i++) {
            const agx_u32 r = order[i];
            if ((seed & 3u) == 2u && rnd() % 50 == 0) continue;            // a row no hit names: declined as well
            const agx_u32 len = 1 + rnd() % maxlen, nhit = 1 + (rnd() % 8 == 0);
            for (agx_u32 e = 0; e < nhit; e++) {
                agx_whit w; memset(&w, 0, sizeof w);
                const bool left2 = rnd() & 1, rev = rnd() & 1, multi = rnd() % 4 == 0, over = rnd() % 40 == 0, other_multi = rnd() % 4 == 0;
                const agx_u32 t0 = over ? n_pos - rnd() % len : (n_pos > len ? rnd() % (n_pos - len + 1) : 0);
                w.row = r; w.len = (agx_u16)len; w.back = (agx_u8)e;
                w.flags = (agx_u8)((left2 ? AGX_WF_LEFT2 : 0) | (rev ? (left2 ? AGX_WF_REV2 : AGX_WF_REV1) : (left2 ? AGX_WF_REV1 : AGX_WF_REV2)));
                // the left mate's runs: pieces of the read with a few bases skipped between them in the read (insertion) or in the reference (deletion); now and then
                // a run that does not fit the read or the unit (the encoder must keep such a row as it is)
                std::vector<agx_wrun> mine;
                if (multi) {
                    agx_u32 q = rnd() % 3, t = t0;
                    const agx_u32 nr = 1 + rnd() % 4;
                    for (agx_u32 k = 0; k < nr && q < len; k++) {
                        const agx_u32 n = 1 + rnd() % (len - q);
                        mine.push_back(agx_wrun{t, (agx_u16)q, (agx_u16)n});
                        q += n; t += n;
                        if (rnd() & 1) q += 1 + rnd() % 3; else t += 1 + rnd() % 5;
                    }
                    if (rnd() % 30 == 0 && !mine.empty()) mine.back().n = (agx_u16)(mine.back().n + 1 + rnd() % 4);      // beyond the read
                } else mine.push_back(agx_wrun{t0, 0, (agx_u16)len});
                agx_wside sd{0, 0, 0};
                if (multi) { (left2 ? sd.runs2 : sd.runs1) = (agx_u32)runs.size(); sd.nruns = left2 ? (agx_u32)mine.size() << 16 : (agx_u32)mine.size(); runs.insert(runs.end(), mine.begin(), mine.end()); w.flags |= left2 ? AGX_WF_RUNS2 : AGX_WF_RUNS1; }
                if (other_multi) { (left2 ? sd.runs1 : sd.runs2) = (agx_u32)runs.size(); sd.nruns |= left2 ? 1u : 1u << 16; runs.push_back(agx_wrun{rnd() % n_pos, 0, 1}); w.flags |= left2 ? AGX_WF_RUNS1 : AGX_WF_RUNS2; }
                (left2 ? w.b : w.a) = t0; (left2 ? w.a : w.b) = rnd();
                if (multi || other_multi) { agx_u32 &field = (w.flags & AGX_WF_RUNS1) ? w.a : w.b; field = (agx_u32)sides.size(); sides.push_back(sd); }
                if (e == 0) {                                              // the anchor: the row's bases follow its geometry
                    for (agx_u32 j = 0; j < stride; j++) {
                        agx_u32 cls;
                        if (j >= len) cls = dirty_tail ? rnd() & 3u : 0u;
                        else {
                            const agx_u32 q = rev ? len - 1 - j : j;
                            unsigned long long x = ~0ull;
                            for (const agx_wrun &g : mine) if (q >= g.q && q < (agx_u32)g.q + g.n) x = (unsigned long long)g.t + (q - g.q);
                            cls = x < n_pos ? (agx_ref_code((agx_u8)ref[x]) & 3u) : rnd() & 3u;      // (inserted and clipped bases: anything)
                            if (rev && x < n_pos) cls = 3u - cls;
                            if (rnd() % 1000 < mut_permille) cls = rnd() & 3u;
                        }
                        codes[(size_t)r * row_bytes + j / 4] |= (agx_u8)(cls << (2 * (j & 3)));
                    }
                }
                hits.push_back(w);
            }
        }
        RowDiffs D;
        if (!build_row_diffs(hits.data(), hits.size(), sides.data(), sides.size(), runs.data(), runs.size(), codes.data(), n_rows, stride, wref.data(), n_pos, threads, D)) { say("build_row_diffs declined"); return 3; }
        if (n_units) *n_units = D.n_units;
        if (n_explicit) *n_explicit = D.n_explicit;
        // as agx_k_expand_rows does it: a block of 64 rows, each row's anchor by counting anchor bits, its units by adding up the counts in front of it, the whole row decoded
        {   std::vector<agx_u8> out(stride + 16);
            for (agx_u32 row = 0; row < n_rows; row++) {
                const agx_u32 blk = row >> 6, h = agx_anchor_select(D.anchor_bits.data(), D.block_first[blk], row & 63u);
                if (h >= hits.size() || hits[h].row != row) { say("row " + std::to_string(row) + ": anchor_select names hit " + std::to_string(h)); return 7; }
                agx_u32 off = D.block_off[blk];
                for (agx_u32 r = row & ~63u; r < row; r++) off += agx_row_units(D.cnt[r], stride);
                const agx_whit w = hits[h]; const agx_u32 cnt = D.cnt[row];
                const agx_wrun *left = nullptr; agx_u32 nruns = 0;
                if (cnt != AGX_ROW_EXPLICIT && !agx_whit_left_simple(w)) { const agx_wside sd = sides[agx_whit_side(w)]; left = runs.data() + agx_wside_left_first(w, sd); nruns = agx_wside_left_count(w, sd); }
                std::fill(out.begin(), out.end(), (agx_u8)0xEE);
                agx_row_decode(wref.data(), D.units.data() + off, cnt, w, left, nruns, stride, out.data());
                for (agx_u32 j = 0; j < stride; j++) {
                    const agx_u8 want = agx_class_vote_code((codes[(size_t)row * row_bytes + j / 4] >> (2 * (j & 3))) & 3u);
                    if (out[j] != want) { say("row " + std::to_string(row) + " base " + std::to_string(j) + ": vote code " + std::to_string(out[j]) + " for " + std::to_string(want) + " (count byte " + std::to_string(cnt) + ")"); return 8; }
                }
                if (out[stride] != 0xEE) { say("row " + std::to_string(row) + " was written beyond its stride"); return 9; }
            }
        }
        std::vector<agx_u8> seen(n_rows, 0);
        for (size_t h = 0; h < hits.size(); h++) {
            if (!((D.anchor_bits[h >> 5] >> (h & 31)) & 1u)) continue;
            const agx_whit w = hits[h];
            const agx_u32 row = w.row;
            if (seen[row]) { say("two anchors for row " + std::to_string(row)); return 4; }
            seen[row] = 1;
            agx_u32 off = D.block_off[row >> 6];
            for (agx_u32 r = row & ~63u; r < row; r++) off += agx_row_units(D.cnt[r], stride);
            const agx_u32 cnt = D.cnt[row];
            if ((size_t)off + agx_row_units(cnt, stride) > D.n_units) { say("row " + std::to_string(row) + " runs past the stream"); return 5; }
            const agx_wrun *left = nullptr; agx_u32 nruns = 0;      // (as agx_k_expand_rows finds them)
            if (cnt != AGX_ROW_EXPLICIT && !agx_whit_left_simple(w)) { const agx_wside sd = sides[agx_whit_side(w)]; left = runs.data() + agx_wside_left_first(w, sd); nruns = agx_wside_left_count(w, sd); }
            for (agx_u32 j0 = 0; j0 < stride; j0 += 16) {
                const agx_u32 c = agx_row_chunk16(wref.data(), D.units.data() + off, cnt, w, left, nruns, j0, stride);
                for (agx_u32 j = j0; j < std::min(stride, j0 + 16); j++) {
                    const agx_u32 want = (codes[(size_t)row * row_bytes + j / 4] >> (2 * (j & 3))) & 3u, got = (c >> (2 * (j - j0))) & 3u;
                    if (want != got) { say("row " + std::to_string(row) + " base " + std::to_string(j) + ": " + std::to_string(got) + " for " + std::to_string(want) + " (count byte " + std::to_string(cnt) + ")"); return 1; }
                }
            }
        }
        for (const agx_whit &w : hits) if (!seen[w.row]) { say("row " + std::to_string(w.row) + " has hits but no anchor"); return 6; }
        return 0;
    } catch (const Error &e) { say(e.msg); return 100 + e.code; }
    catch (const std::exception &e) { say(e.what()); return 99; }
}

// The same on a unit's real files: general loaders -> stage_pairs -> pack_reference -> build_row_diffs, every row decoded the device's way and compared with its
// 2-bit classes; out[0..3] = rows, rows kept as they are, bytes of the 2-bit rows, bytes of the upload form.  -2: the unit sequence does not pack (soft-masked).
extern "C" int agx_hostsim_rowdiff_unit(const char *tmp_dir, int unit, int k, long batch, int threads, unsigned long long *out, char *msg, size_t msg_len) {
    auto say = [&](const std::string &m) { if (msg && msg_len) snprintf(msg, msg_len, "%s", m.c_str()); };
    say("");
    try {
        const std::string d = tmp_dir, s = std::to_string(unit);
        struct VSink : StageSink { std::vector<std::vector<char>> keep; void *take(int, size_t bytes) override { keep.emplace_back(bytes + 64); return keep.back().data(); } } A;
        Threads T; Pairs P; StagedPairs S;
        load_unit_reference(d + "/_genome." + s + ".fa", T.ref);
        thread_contigs_from_files(d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", T);      // (appends positions to the unit sequence)
        load_pairs_from_files(d + "/_reads.fa", d + "/_reads_genome." + s + ".bowtie", batch, (agx_u32)k, P, nullptr);
        stage_pairs(P, (agx_u32)k, (unsigned)threads, A, S);
        const size_t n_pos = T.ref.size();
        std::vector<agx_u8> packed((n_pos + 3) / 4 + 64, 0x5A);
        std::vector<agx_refx> others;
        if (!pack_reference(T.ref.data(), n_pos, 2, packed.data(), others)) return -2;
        std::vector<agx_u32> wref(packed.size() / 4 + 1); memcpy(wref.data(), packed.data(), packed.size());
        RowDiffs D;
        if (!build_row_diffs(S.hits, S.nh, S.sides, S.n_sides, S.runs, S.n_runs, S.codes, S.n_rows, S.stride, wref.data(), n_pos, (unsigned)threads, D)) { say("build_row_diffs declined"); return 3; }
        out[0] = S.n_rows; out[1] = D.n_explicit; out[2] = S.n_codes; out[3] = D.n_units * 2 + D.cnt.size() + (D.block_off.size() + D.block_first.size() + D.anchor_bits.size()) * 4;
        {   std::vector<agx_u8> o(S.stride + 16);
            for (agx_u32 row = 0; row < S.n_rows; row++) {
                const agx_u32 blk = row >> 6, h = agx_anchor_select(D.anchor_bits.data(), D.block_first[blk], row & 63u);
                if (h >= S.nh || S.hits[h].row != row) { say("row " + std::to_string(row) + ": anchor_select names hit " + std::to_string(h)); return 7; }
                agx_u32 off = D.block_off[blk];
                for (agx_u32 r = row & ~63u; r < row; r++) off += agx_row_units(D.cnt[r], S.stride);
                const agx_whit w = S.hits[h]; const agx_u32 cnt = D.cnt[row];
                const agx_wrun *left = nullptr; agx_u32 nruns = 0;
                if (cnt != AGX_ROW_EXPLICIT && !agx_whit_left_simple(w)) { const agx_wside sd = S.sides[agx_whit_side(w)]; left = S.runs + agx_wside_left_first(w, sd); nruns = agx_wside_left_count(w, sd); }
                agx_row_decode(wref.data(), D.units.data() + off, cnt, w, left, nruns, S.stride, o.data());
                for (agx_u32 j = 0; j < S.stride; j++) {
                    const agx_u8 want = agx_class_vote_code((S.codes[(size_t)row * (S.stride / 4) + j / 4] >> (2 * (j & 3))) & 3u);
                    if (o[j] != want) { say("row " + std::to_string(row) + " base " + std::to_string(j) + ": vote code " + std::to_string(o[j]) + " for " + std::to_string(want)); return 8; }
                }
            }
        }
        const size_t row_bytes = S.stride / 4;
        std::vector<agx_u8> seen(S.n_rows, 0);
        for (size_t h = 0; h < S.nh; h++) {
            if (!((D.anchor_bits[h >> 5] >> (h & 31)) & 1u)) continue;
            const agx_whit w = S.hits[h]; const agx_u32 row = w.row;
            if (seen[row]) { say("two anchors for row " + std::to_string(row)); return 4; }
            seen[row] = 1;
            agx_u32 off = D.block_off[row >> 6];
            for (agx_u32 r = row & ~63u; r < row; r++) off += agx_row_units(D.cnt[r], S.stride);
            const agx_wrun *left = nullptr; agx_u32 nruns = 0;
            if (D.cnt[row] != AGX_ROW_EXPLICIT && !agx_whit_left_simple(w)) { const agx_wside sd = S.sides[agx_whit_side(w)]; left = S.runs + agx_wside_left_first(w, sd); nruns = agx_wside_left_count(w, sd); }
            for (agx_u32 j0 = 0; j0 < S.stride; j0 += 16) {
                const agx_u32 c = agx_row_chunk16(wref.data(), D.units.data() + off, D.cnt[row], w, left, nruns, j0, S.stride);
                agx_u32 want = 0; memcpy(&want, S.codes + (size_t)row * row_bytes + j0 / 4, std::min<size_t>(4, row_bytes - j0 / 4));
                if (c != want) { say("row " + std::to_string(row) + " bases " + std::to_string(j0) + "..: " + std::to_string(c) + " for " + std::to_string(want)); return 1; }
            }
        }
        for (size_t r = 0; r < S.n_rows; r++) if (!seen[r]) { say("row " + std::to_string(r) + " has no anchor"); return 6; }
        return 0;
    } catch (const Error &e) { say(e.msg); return 100 + e.code; }
    catch (const std::exception &e) { say(e.what()); return 99; }
}
