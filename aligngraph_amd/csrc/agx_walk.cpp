// agx_walk.cpp — coverage-pruned path walk, contig join and mate-guided scaffolding on the packed graph.
//
// Follows extdContigs1 (AG:1954-2204), extdContigs2 (AG:2296-2380) and scaffoldContigs (AG:2396-2464) of
// /root/reference/AlignGraph/AlignGraph.cpp.  The walk is sequential BY SPECIFICATION: every branch decision
// depends on which nodes earlier walks already consumed (AG:2027-2033, 2096-2113), the position scan skips ahead
// inside long records (AG:2194-2202) and a record is suppressed against the previously WRITTEN one (AG:2176).
// It therefore runs on the host over the flat node table the kernels produced: prune flags and consensus bases
// were fused into the node sweep, so the walk touches 1 byte of state per node plus its out-edge slots.
#include "agx_host.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace agx {
namespace {

struct Rec {                 // Contig, AG:123-139
    int extended;
    agx_u32 sID, sOff, eID, eOff, sID0, sOff0, eID0, eOff0;
    std::string nuc;
};

inline bool contains(agx_u32 sID1, agx_u32 sOff1, agx_u32 eID1, agx_u32 eOff1, agx_u32 sID2, agx_u32 sOff2, agx_u32 eID2, agx_u32 eOff2) {
    return sID1 == sID2 && eID1 == eID2 && sOff1 <= sOff2 && eOff1 >= eOff2;      // AG:1897-1902
}

inline void fasta_body(std::string &out, const char *s, size_t n) {
    for (size_t i = 0; i < n; i += 60) { const size_t m = n - i < 60 ? n - i : 60; out.append(s + i, m); out.push_back('\n'); }
}

struct Walker {
    const Threads &T; const Pairs &P; const GraphView &G;
    std::vector<agx_u8> done;                       // traversed flag per node, seeded with the prune result
    std::vector<agx_edge_ovf> ovf;                  // sorted, unique
    Walker(const Threads &t, const Pairs &p, const GraphView &g) : T(t), P(p), G(g), done(g.n_nodes) {
        for (agx_u32 i = 0; i < g.n_nodes; i++) done[i] = (g.flags[i] & AGX_NF_DEAD) ? 1 : 0;
        ovf.assign(g.ovf, g.ovf + g.n_ovf);
        std::sort(ovf.begin(), ovf.end(), [](const agx_edge_ovf &a, const agx_edge_ovf &b) { return a.src != b.src ? a.src < b.src : a.dst < b.dst; });
        ovf.erase(std::unique(ovf.begin(), ovf.end(), [](const agx_edge_ovf &a, const agx_edge_ovf &b) { return a.src == b.src && a.dst == b.dst; }), ovf.end());
    }
    // the single live successor of node v, if it has exactly one (AG:2020-2033); returns the count, capped at 2
    int live_successors(agx_u32 v, agx_u32 &target) const {
        int n = 0;
        const agx_u32 *s = G.next + (size_t)v * AGX_MAXE;
        for (agx_u32 e = 0; e < AGX_MAXE && s[e] != AGX_NONE; e++) if (!done[s[e]]) { target = s[e]; if (++n > 1) return n; }
        if (G.flags[v] & AGX_NF_EOVF) {
            auto it = std::lower_bound(ovf.begin(), ovf.end(), v, [](const agx_edge_ovf &a, agx_u32 key) { return a.src < key; });
            for (; it != ovf.end() && it->src == v; ++it) {
                bool inl = false; for (agx_u32 e = 0; e < AGX_MAXE; e++) inl |= s[e] == it->dst;
                if (!inl && !done[it->dst]) { target = it->dst; if (++n > 1) return n; }
            }
        }
        return n;
    }
    // k-mer string of node v from its read reference (agx_sref)
    void kmer_string(agx_u32 v, std::string &out) const {
        const agx_sref r = G.sref[v];
        const agx_u32 first = r.qlen & 0xFFFFu, len = (r.qlen >> 16) & 0x7FFFu; const bool rev = (r.qlen >> 31) != 0;
        out.clear();
        const char *p = P.bases.data() + (size_t)r.slot * P.stride;
        for (agx_u32 i = 0; i < len; i++) {
            if (!rev) out.push_back(p[first + i]);
            else { const char c = p[first - i]; out.push_back(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c); }
        }
    }
};

std::string header_of(agx_u32 id, const Rec &c) {      // AG:2178
    char buf[256];
    std::snprintf(buf, sizeof buf, ">%u, %d, %u, %u, %u, %u, %u, %u, %u, %u \n", id, c.extended, c.sID, c.sOff, c.eID, c.eOff, c.sID0, c.sOff0, c.eID0, c.eOff0);
    return buf;
}

// extdContigs1, AG:1954-2204
void walk(Walker &W, std::string &pre_out, std::vector<Rec> &written) {
    const GraphView &G = W.G; const Threads &T = W.T;
    agx_u32 seqID = 0, sIDBak = AGX_NONE, sOffBak = AGX_NONE, eIDBak = AGX_NONE, eOffBak = AGX_NONE;
    agx_u32 pos_bak = 0;                         // cppBak of the reference (function scope)
    std::string kmer;
    for (agx_u32 cp = 0; cp < G.n_pos;) {
        const agx_u32 n0 = G.node_start[cp], nc = G.node_cnt[cp];
        for (agx_u32 ip = 0; ip < nc; ip++) {
            if (W.done[n0 + ip]) continue;
            Rec C; C.sID = 0; C.sOff = cp; C.extended = 0;
            agx_u32 cur = n0 + ip;               // current k-mer node (mode 1)
            C.sID0 = G.off0[cur] == AGX_NONE ? AGX_NONE : 0; C.sOff0 = G.off0[cur];
            agx_u32 cpp = cp, ipp = ip; int mode = 1;      // mode = kMerTag
            agx_u32 last = cur;
            while ((mode == 1 && !W.done[cur]) || mode == 0) {
                if (mode == 0) {                            // on a conti-mer, AG:2061-2138
                    // follow the conti-mer chain to its end in one append (the reference steps through it one base at a time)
                    const agx_u32 ci = T.cm_start[cpp] + ipp, ch = T.cm_chain[ci];
                    const size_t from = T.chain_off[ch] + T.cm_idx[ci], to = T.chain_off[ch + 1];
                    C.nuc.append(T.chain_str, from, to - from); C.extended = 1;
                    if (to - from > 1) { pos_bak = T.chain_end_pos[ch]; cpp = pos_bak; }
                    {
                        // chain end: hop onto the k-mer graph only through the single live node here and its single live edge (AG:2093-2136)
                        const agx_u32 b0 = G.node_start[cpp], bn = G.node_cnt[cpp];
                        agx_u32 live = 0, item = 0;
                        for (agx_u32 v = 0; v < bn; v++) if (!W.done[b0 + v]) { live++; item = b0 + v; }
                        agx_u32 tgt = 0; int ns = 0;
                        if (live == 1) ns = W.live_successors(item, tgt);
                        if (ns == 1) { cur = tgt; pos_bak = G.xpos[tgt]; cpp = pos_bak; mode = W.done[cur] ? -2 : 1; }
                        else mode = -2;
                    }
                } else {                                    // on a k-mer node, AG:1995-2060
                    const char c = (char)G.base[cur];
                    C.nuc.push_back(c != 'X' ? c : T.ref[cpp]);
                    if (G.flags[cur] & AGX_NF_CONTIG) C.extended = 1;
                    W.done[cur] = 1; last = cur;
                    agx_u32 tgt = 0;
                    const int ns = W.live_successors(cur, tgt);
                    if (ns == 1) { cur = tgt; pos_bak = G.xpos[tgt]; cpp = pos_bak; }
                    else {
                        const agx_u32 c0 = T.cm_start[cpp], cn = T.cm_start[cpp + 1] - c0;
                        if (cn == 1 && T.cm[c0].next_off != AGX_NONE) { pos_bak = T.cm[c0].next_off; ipp = T.cm[c0].next_item; cpp = pos_bak; mode = 0; }
                        else mode = -1;
                    }
                }
            }
            // end bookkeeping, AG:2142-2173
            C.eID = 0; C.eOff = mode == 1 ? pos_bak : cpp;
            if (mode == 1 || mode == -1) {
                C.eID0 = G.off0[cur] == AGX_NONE ? AGX_NONE : 0; C.eOff0 = G.off0[cur];
                W.kmer_string(last, kmer);
                if (kmer.size() > 1) C.nuc.append(kmer, 1, std::string::npos);
                C.eOff = C.eOff + (agx_u32)kmer.size() - 1; C.eOff0 = C.eOff0 + (agx_u32)kmer.size() - 1;
            } else { C.eID0 = AGX_NONE; C.eOff0 = AGX_NONE; }
            if (!contains(sIDBak, sOffBak, eIDBak, eOffBak, C.sID, C.sOff, C.eID, C.eOff)) {        // AG:2176-2189
                pre_out += header_of(seqID++, C);
                fasta_body(pre_out, C.nuc.data(), C.nuc.size());
                sIDBak = C.sID; sOffBak = C.sOff; eIDBak = C.eID; eOffBak = C.eOff;
                written.push_back(std::move(C));
            }
        }
        if (eOffBak - sOffBak > 100000 && eIDBak == 0 && cp + 1000 < eOffBak) cp += 1000; else cp++;     // AG:2194-2202
    }
}

}  // namespace

namespace {

// extdContigs2, AG:2296-2380
void join(std::vector<Rec> &c) {
    const int n = (int)c.size();
    for (int cp = 0; cp < n; cp++) if (c[cp].extended == 1)
        for (int q = cp + 1; q < n; q++) {
            if (contains(c[cp].sID, c[cp].sOff, c[cp].eID, c[cp].eOff, c[q].sID, c[q].sOff, c[q].eID, c[q].eOff)) c[q].extended = 2;
            else if (c[cp].eID != c[q].sID || c[cp].eOff < c[q].sOff) break;
        }
    for (int cp = n - 1; cp >= 0; cp--) if (c[cp].extended == 1)
        for (int q = cp - 1; q >= 0; q--) {
            if (contains(c[cp].sID, c[cp].sOff, c[cp].eID, c[cp].eOff, c[q].sID, c[q].sOff, c[q].eID, c[q].eOff)) c[q].extended = 2;
            else if (c[q].eID != c[cp].sID || c[q].eOff < c[cp].sOff) break;
        }
    for (int cp = 0; cp < n; cp++) {
        while (c[cp].extended == 1) {
            int cand = -1, ncand = 0;
            for (int q = cp + 1; q < n; q++) if (c[q].extended != 2) { if (c[cp].eOff >= c[q].sOff) { cand = q; ncand++; } else break; }
            if (ncand != 1) break;
            Rec &d = c[cand]; d.extended = 2;
            const int from = (int)(c[cp].eOff - d.sOff + 1);
            for (size_t np = (size_t)from; np < d.nuc.size(); np++) c[cp].nuc.push_back(d.nuc[np]);
            c[cp].eID = d.eID; c[cp].eOff = d.eOff; c[cp].eID0 = d.eID0; c[cp].eOff0 = d.eOff0;
        }
    }
}

inline int overlaps(agx_u32 x1, agx_u32 y1, agx_u32 x2, agx_u32 y2) {      // AG:2388-2394
    return (x1 <= x2 && x2 <= y1 && y1 <= y2 && (int)y1 - (int)x2 > 0) || (x2 <= x1 && x1 <= y2 && y2 <= y1 && (int)y2 - (int)x1 > 0) ||
           (x1 <= x2 && x2 <= y2 && y2 <= y1 && (int)y2 - (int)x2 > 0) || (x2 <= x1 && x1 <= y1 && y1 <= y2 && (int)y1 - (int)x1 > 0);
}

// scaffoldContigs, AG:2396-2464
void scaffold(const Threads &T, const GraphView &G, std::vector<Rec> &c, std::string &out) {
    std::vector<std::string> sc;
    const agx_u32 n = (agx_u32)c.size();
    for (agx_u32 cp = 0; cp < n; cp++) {
        if (!(c[cp].sID != AGX_NONE && c[cp].extended == 1)) continue;
        sc.push_back(c[cp].nuc); c[cp].sID = AGX_NONE;
        bool cont = true;
        while (c[cp].sID0 == c[cp].eID0 && cont) {
            cont = false;
            for (agx_u32 q = cp + 1; q < n; q++) {
                if (!(c[cp].eID0 == c[q].sID && c[q].sID == c[q].eID && overlaps(c[cp].sOff0, c[cp].eOff0, c[q].sOff, c[q].eOff) && c[q].extended == 1)) continue;
                if (c[q].sOff > c[cp].eOff) {
                    const agx_u32 gap = c[q].sOff - c[cp].eOff - 1; agx_u32 covered = 0;
                    for (agx_u32 i = 0; i < gap; i++) { const agx_u32 x = c[cp].eOff + i + 1; if (G.node_cnt[x] > 0 || T.cm_start[x + 1] > T.cm_start[x]) covered++; }
                    if (gap == 0 || (double)(int)covered / gap >= 0.5) sc.back().append(T.ref, c[cp].eOff + 1, gap);
                    else continue;
                }
                sc.back() += c[q].nuc; c[q].sID = AGX_NONE; cp = q; cont = true;
                break;
            }
        }
    }
    for (size_t i = 0; i < sc.size(); i++) { out += ">" + std::to_string(i) + "\n"; fasta_body(out, sc[i].data(), sc[i].size()); }
}

}  // namespace

// chains of conti-mers: heads are the conti-mers nobody links to
void build_chains(Threads &T) {
    const size_t n = T.cm.size(), n_pos = T.ref.size();
    T.cm_chain.assign(n, AGX_NONE); T.cm_idx.assign(n, 0); T.chain_off.assign(1, 0); T.chain_end_pos.clear(); T.chain_str.clear();
    T.chain_str.reserve(n);
    std::vector<agx_u8> linked(n, 0);
    auto index_of = [&](agx_u32 off, agx_u32 item) -> size_t {
        if (off >= n_pos || T.cm_start[off] + item >= T.cm_start[off + 1]) throw Error{E_ARG, "conti-mer link to a missing entry"};
        return (size_t)T.cm_start[off] + item;
    };
    for (size_t i = 0; i < n; i++) if (T.cm[i].next_off != AGX_NONE) linked[index_of(T.cm[i].next_off, T.cm[i].next_item)] = 1;
    std::vector<agx_u32> pos_of(n);
    for (size_t x = 0; x < n_pos; x++) for (agx_u32 i = T.cm_start[x]; i < T.cm_start[x + 1]; i++) pos_of[i] = (agx_u32)x;
    for (size_t h = 0; h < n; h++) {
        if (linked[h]) continue;
        const agx_u32 ch = (agx_u32)T.chain_end_pos.size();
        size_t c = h; agx_u32 idx = 0;
        for (;;) {
            if (T.cm_chain[c] != AGX_NONE) throw Error{E_ARG, "conti-mer chains are not simple lists"};
            T.cm_chain[c] = ch; T.cm_idx[c] = idx++; T.chain_str.push_back(T.cm[c].nuc);
            if (T.cm[c].next_off == AGX_NONE) break;
            c = index_of(T.cm[c].next_off, T.cm[c].next_item);
        }
        T.chain_end_pos.push_back(pos_of[c]); T.chain_off.push_back(T.chain_str.size());
    }
    for (size_t i = 0; i < n; i++) if (T.cm_chain[i] == AGX_NONE) throw Error{E_ARG, "conti-mer cycle"};
}

void walk_join_scaffold(const Threads &T, const Pairs &P, const GraphView &G, UnitOutput &out) {
    if (T.cm_chain.size() != T.cm.size()) throw Error{E_ARG, "conti-mer chains were not built"};
    Walker W(T, P, G);
    std::vector<Rec> recs;
    out.initial_contigs = T.initial_contigs;
    out.pre_extended.clear(); out.extended.clear();
    const bool timing = getenv("AGX_WALK_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    walk(W, out.pre_extended, recs);
    double t1 = now();
    join(recs);
    double t2 = now();
    scaffold(T, G, recs, out.extended);
    double t3 = now();
    if (timing) fprintf(stderr, "[agx walk] walk %.1f ms, join %.1f ms, scaffold %.1f ms, %zu records\n", t1 - t0, t2 - t1, t3 - t2, recs.size());
}

}  // namespace agx
