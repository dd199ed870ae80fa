// Host-side tile planner for the fused spatial-layer kernel (integer bookkeeping only).
//
// A "side" of a drainage network is (ADJ, INC): the GAT pattern among its primary rows and the
// incidence support that feeds them from the secondary rows:
//   node side : primary = nodes, ADJ = adj  (N x N), secondary = links, INC = inc_n (N x E)
//   link side : primary = links, ADJ = edge_adj (E x E), secondary = nodes, INC = inc_e (E x N)
// (reference: emulator.py:227-230 -- NodeEdge aggregation then GAT, on nodes and on links).
//
// Primary rows are clustered into tiles of at most t_max rows by a bottom-up packing of a BFS
// spanning forest of ADJ: drainage networks are near-trees, so a cluster of T rows has O(1) cut
// links instead of the O(T) a contiguous id range would have.  One workgroup computes one tile:
//   own  rows  : outputs it writes                                   (first n_own of prim)
//   halo rows  : ADJ-neighbours of own rows outside the cluster -- their transformed features are
//                recomputed locally instead of exchanged            (rest of prim)
//   sec  rows  : every secondary row incident to a prim row (their MLP output is recomputed too)
// All lists are int32; local CSR indices point into prim / sec.  Tested in tests/test_tile_plan.py.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

namespace uds {

struct HostCsr {
  int64_t n_rows = 0, n_cols = 0;
  std::vector<int32_t> rowptr, col;
};

constexpr int TILE_HDR_INTS = 8;  // n_own, n_prim, n_sec, n_inc, n_adj, pool_off, side, meta_len
// ints of a tile's fixed-stride block: header + metadata, padded to 16 bytes (k_fused_tile fetches it whole)
inline int tile_block_ints(int meta_cap) { return (TILE_HDR_INTS + meta_cap + 3) & ~3; }

struct SidePlan {
  int n_tiles = 0;
  int p_cap = 0, q_cap = 0, meta_cap = 0;  // maxima over tiles (rows rounded up to 16, meta to 4 ints)
  std::vector<int32_t> hdr;                // n_tiles * TILE_HDR_INTS
  std::vector<int32_t> pool;               // per tile: prim[n_prim] sec[n_sec] inc_ptr[n_prim+1] inc_loc[n_inc]
                                           //           inc_w[n_inc] adj_ptr[n_own+1] adj_loc[n_adj]
  std::vector<int32_t> key;                // locality key per tile (median BFS index of the nodes it touches)
  std::vector<int32_t> bfs_index;          // BFS visiting index of every primary row
};

// BFS spanning forest of adj (self loops ignored) + bottom-up packing into clusters of <= t_max rows.
inline void cluster_rows(const HostCsr &adj, int t_max, std::vector<int32_t> &cluster_of, int &n_clusters,
                         std::vector<int32_t> &bfs_index) {
  const int32_t n = (int32_t)adj.n_rows;
  cluster_of.assign(n, -1);
  bfs_index.assign(n, -1);
  std::vector<int32_t> order, parent(n, -1);
  order.reserve(n);
  for (int32_t root = 0; root < n; ++root) {
    if (bfs_index[root] >= 0) continue;
    bfs_index[root] = (int32_t)order.size();
    order.push_back(root);
    for (size_t head = order.size() - 1; head < order.size(); ++head) {
      const int32_t v = order[head];
      for (int32_t p = adj.rowptr[v]; p < adj.rowptr[v + 1]; ++p) {
        const int32_t u = adj.col[p];
        if (bfs_index[u] < 0) {
          bfs_index[u] = (int32_t)order.size();
          parent[u] = v;
          order.push_back(u);
        }
      }
    }
  }
  n_clusters = 0;
  const int t_min = std::max(1, (3 * t_max) / 4);
  std::vector<std::vector<int32_t>> open(n);          // rows of the still-open subtree hanging at v
  std::vector<std::vector<int32_t>> kids(n);          // children whose open list is non-empty
  auto close = [&](const std::vector<int32_t> &rows) {
    if (rows.empty()) return;
    for (int32_t r : rows) cluster_of[r] = n_clusters;
    ++n_clusters;
  };
  for (int64_t k = (int64_t)order.size() - 1; k >= 0; --k) {
    const int32_t v = order[k];
    std::vector<int32_t> &mine = open[v];
    mine.push_back(v);
    std::vector<int32_t> &ch = kids[v];
    std::stable_sort(ch.begin(), ch.end(), [&](int32_t x, int32_t y) { return open[x].size() > open[y].size(); });
    std::vector<int32_t> spill;
    for (int32_t c : ch) {
      std::vector<int32_t> &lst = open[c];
      if ((int)(mine.size() + lst.size()) <= t_max) {
        mine.insert(mine.end(), lst.begin(), lst.end());
      } else {
        if ((int)(spill.size() + lst.size()) > t_max) {
          close(spill);
          spill.clear();
        }
        spill.insert(spill.end(), lst.begin(), lst.end());
      }
      std::vector<int32_t>().swap(lst);
    }
    close(spill);
    if ((int)mine.size() >= t_min || parent[v] < 0) {
      close(mine);
      std::vector<int32_t>().swap(mine);
    } else {
      kids[parent[v]].push_back(v);
    }
    std::vector<int32_t>().swap(ch);
  }
}

// node_bfs: BFS index of every NODE (for the locality key).  For the node side key_cols == nullptr
// (keys come from the primary rows themselves); for the link side the key of a link is the smaller
// node_bfs of its endpoints (the cols of its INC row).
// Rows a tile of `own` rows would stage: primary (own + outside ADJ-neighbours) and secondary (INC-incident) counts.
inline void tile_footprint(const HostCsr &adj, const HostCsr &inc, const std::vector<int32_t> &own, std::vector<int32_t> &mark_p,
                           std::vector<int32_t> &mark_s, int &n_prim, int &n_sec) {
  std::vector<int32_t> prim, sec;
  for (int32_t r : own)
    for (int32_t p = adj.rowptr[r]; p < adj.rowptr[r + 1]; ++p)
      if (!mark_p[adj.col[p]]) {
        mark_p[adj.col[p]] = 1;
        prim.push_back(adj.col[p]);
      }
  for (int32_t r : own)
    if (!mark_p[r]) {   // a row always counts itself (patterns normally hold the self loop already)
      mark_p[r] = 1;
      prim.push_back(r);
    }
  for (int32_t r : prim)
    for (int32_t p = inc.rowptr[r]; p < inc.rowptr[r + 1]; ++p)
      if (!mark_s[inc.col[p]]) {
        mark_s[inc.col[p]] = 1;
        sec.push_back(inc.col[p]);
      }
  n_prim = (int)prim.size();
  n_sec = (int)sec.size();
  for (int32_t r : prim) mark_p[r] = 0;
  for (int32_t q : sec) mark_s[q] = 0;
}

// Greedy agglomeration of small clusters into tiles that FILL the footprint limits.  A tile costs a workgroup about
// the same time whether it is half empty or full (its phases are one 16-row block per wave either way), so what counts
// is the number of tiles.  Seeds are taken in BFS order; a seed keeps absorbing the adjacent cluster it shares the most
// ADJ entries with, as long as the union still has <= own_limit own rows, <= p_limit primary rows (own + halo) and
// <= q_limit secondary rows.  Merging neighbours also turns their mutual halo rows into own rows.
inline void merge_clusters(const HostCsr &adj, const HostCsr &inc, const std::vector<int32_t> &bfs_index, int own_limit, int p_limit,
                           int q_limit, std::vector<std::vector<int32_t>> &members) {
  const int32_t n = (int32_t)adj.n_rows;
  const int nc = (int)members.size();
  std::vector<int32_t> cluster_of(n, -1);
  for (int c = 0; c < nc; ++c)
    for (int32_t r : members[c]) cluster_of[r] = c;
  std::vector<int32_t> key(nc, INT32_MAX);
  for (int c = 0; c < nc; ++c)
    for (int32_t r : members[c]) key[c] = std::min(key[c], bfs_index[r]);
  std::vector<int32_t> order(nc);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return key[x] < key[y]; });
  std::vector<char> alive(nc, 1), done(nc, 0);
  std::vector<int32_t> mark_p(n, 0), mark_s((size_t)inc.n_cols, 0), shared(nc, 0), touched;
  std::vector<int32_t> trial;
  for (int32_t seed : order) {
    if (!alive[seed] || done[seed]) continue;
    for (;;) {
      // clusters adjacent to the seed and the number of ADJ entries that cross
      touched.clear();
      for (int32_t r : members[seed])
        for (int32_t p = adj.rowptr[r]; p < adj.rowptr[r + 1]; ++p) {
          const int32_t c = cluster_of[adj.col[p]];
          if (c == seed || done[c]) continue;
          if (shared[c]++ == 0) touched.push_back(c);
        }
      std::stable_sort(touched.begin(), touched.end(), [&](int32_t x, int32_t y) { return shared[x] > shared[y]; });
      int32_t pick = -1;
      for (int32_t c : touched) {
        if ((int)(members[seed].size() + members[c].size()) > own_limit) continue;
        trial.assign(members[seed].begin(), members[seed].end());
        trial.insert(trial.end(), members[c].begin(), members[c].end());
        int np_ = 0, nq_ = 0;
        tile_footprint(adj, inc, trial, mark_p, mark_s, np_, nq_);
        if (np_ <= p_limit && nq_ <= q_limit) {
          pick = c;
          break;
        }
      }
      for (int32_t c : touched) shared[c] = 0;
      if (pick < 0) break;
      for (int32_t r : members[pick]) cluster_of[r] = seed;
      members[seed].insert(members[seed].end(), members[pick].begin(), members[pick].end());
      std::vector<int32_t>().swap(members[pick]);
      alive[pick] = 0;
    }
    done[seed] = 1;
  }
  // Leftovers: pockets the grown tiles closed in.  A tile need not be connected (footprints of distant pieces simply
  // add), so they are bin-packed -- largest first, first fit among the most recently opened bins, exact footprint.
  std::vector<std::vector<int32_t>> out, small;
  for (int32_t c : order)
    if (alive[c] && !members[c].empty()) {
      if ((int)members[c].size() * 4 >= own_limit * 3) out.push_back(std::move(members[c]));
      else small.push_back(std::move(members[c]));
    }
  std::stable_sort(small.begin(), small.end(), [](const std::vector<int32_t> &x, const std::vector<int32_t> &y) { return x.size() > y.size(); });
  std::vector<std::vector<int32_t>> bins;
  constexpr size_t WINDOW = 48;
  for (auto &piece : small) {
    bool placed = false;
    const size_t first = bins.size() > WINDOW ? bins.size() - WINDOW : 0;
    for (size_t b = first; b < bins.size() && !placed; ++b) {
      if ((int)(bins[b].size() + piece.size()) > own_limit) continue;
      trial.assign(bins[b].begin(), bins[b].end());
      trial.insert(trial.end(), piece.begin(), piece.end());
      int np_ = 0, nq_ = 0;
      tile_footprint(adj, inc, trial, mark_p, mark_s, np_, nq_);
      if (np_ <= p_limit && nq_ <= q_limit) {
        bins[b].swap(trial);
        placed = true;
      }
    }
    if (!placed) bins.push_back(std::move(piece));
  }
  for (auto &b : bins) out.push_back(std::move(b));
  members.swap(out);
}

// p_limit / q_limit (0 = none): footprint limits of a tile (primary rows incl. halo / secondary rows).  With limits the
// rows are first packed into small subtree clusters (t_max / 6 rows) which merge_clusters() then agglomerates into
// tiles of at most t_max own rows that fill the limits; a cluster that still exceeds them (a hub row) is bisected in BFS
// order until it fits, so one outlier does not set the LDS size of every workgroup.
inline SidePlan build_side_plan(const HostCsr &adj, const HostCsr &inc, int t_max, int side,
                                const std::vector<int32_t> *node_bfs, int p_limit = 0, int q_limit = 0) {
  SidePlan plan;
  std::vector<int32_t> cluster_of;
  int n_clusters = 0;
  const bool limits = p_limit > 0 && q_limit > 0;
  cluster_rows(adj, limits ? std::max(4, t_max / 6) : t_max, cluster_of, n_clusters, plan.bfs_index);
  const int32_t n = (int32_t)adj.n_rows;
  std::vector<std::vector<int32_t>> members(n_clusters);
  for (int32_t r = 0; r < n; ++r) members[cluster_of[r]].push_back(r);   // ascending ids
  if (p_limit > 0 || q_limit > 0) {
    std::vector<int32_t> mark_p(n, 0), mark_s((size_t)inc.n_cols, 0);
    std::vector<std::vector<int32_t>> fitted;
    std::vector<std::vector<int32_t>> work(members.rbegin(), members.rend());
    while (!work.empty()) {
      std::vector<int32_t> c = std::move(work.back());
      work.pop_back();
      int np_ = 0, nq_ = 0;
      tile_footprint(adj, inc, c, mark_p, mark_s, np_, nq_);
      const bool too_big = (p_limit > 0 && np_ > p_limit) || (q_limit > 0 && nq_ > q_limit);
      if (!too_big || c.size() <= 1) {
        fitted.push_back(std::move(c));
        continue;
      }
      std::stable_sort(c.begin(), c.end(), [&](int32_t x, int32_t y) { return plan.bfs_index[x] < plan.bfs_index[y]; });
      const size_t half = c.size() / 2;
      work.emplace_back(c.begin() + half, c.end());
      work.emplace_back(c.begin(), c.begin() + half);
    }
    members.swap(fitted);
    if (limits) merge_clusters(adj, inc, plan.bfs_index, t_max, p_limit, q_limit, members);
    n_clusters = (int)members.size();
    for (auto &m : members) std::sort(m.begin(), m.end());
  }
  // degree-sorted schedule inside a tile: descending degree, ties by id (stable), so the 16 rows one wave aggregates
  // together have near-equal neighbour counts
  for (auto &m : members)
    std::stable_sort(m.begin(), m.end(), [&](int32_t x, int32_t y) {
      return (adj.rowptr[x + 1] - adj.rowptr[x]) > (adj.rowptr[y + 1] - adj.rowptr[y]);
    });
  std::vector<int32_t> ploc(n, -1), sloc((size_t)inc.n_cols, -1);
  plan.n_tiles = n_clusters;
  plan.hdr.reserve((size_t)n_clusters * TILE_HDR_INTS);
  for (int c = 0; c < n_clusters; ++c) {
    const std::vector<int32_t> &own = members[c];
    std::vector<int32_t> prim(own), halo, sec;
    for (size_t i = 0; i < own.size(); ++i) ploc[own[i]] = (int32_t)i;
    for (int32_t r : own)
      for (int32_t p = adj.rowptr[r]; p < adj.rowptr[r + 1]; ++p) {
        const int32_t u = adj.col[p];
        if (ploc[u] < 0) {
          ploc[u] = 0;  // mark, real index assigned after sorting
          halo.push_back(u);
        }
      }
    std::sort(halo.begin(), halo.end());
    for (int32_t u : halo) {
      ploc[u] = (int32_t)prim.size();
      prim.push_back(u);
    }
    for (int32_t r : prim)
      for (int32_t p = inc.rowptr[r]; p < inc.rowptr[r + 1]; ++p) {
        const int32_t q = inc.col[p];
        if (sloc[q] < 0) {
          sloc[q] = 0;
          sec.push_back(q);
        }
      }
    std::sort(sec.begin(), sec.end());
    for (size_t i = 0; i < sec.size(); ++i) sloc[sec[i]] = (int32_t)i;

    const int32_t off = (int32_t)plan.pool.size();
    plan.pool.insert(plan.pool.end(), prim.begin(), prim.end());
    plan.pool.insert(plan.pool.end(), sec.begin(), sec.end());
    std::vector<int32_t> inc_loc, inc_w, adj_loc;
    plan.pool.push_back(0);
    for (int32_t r : prim) {
      for (int32_t p = inc.rowptr[r]; p < inc.rowptr[r + 1]; ++p) {
        inc_loc.push_back(sloc[inc.col[p]]);
        inc_w.push_back(p);                       // position in the global support-value array
      }
      plan.pool.push_back((int32_t)inc_loc.size());
    }
    plan.pool.insert(plan.pool.end(), inc_loc.begin(), inc_loc.end());
    plan.pool.insert(plan.pool.end(), inc_w.begin(), inc_w.end());
    plan.pool.push_back(0);
    for (int32_t r : own) {
      for (int32_t p = adj.rowptr[r]; p < adj.rowptr[r + 1]; ++p) adj_loc.push_back(ploc[adj.col[p]]);
      plan.pool.push_back((int32_t)adj_loc.size());
    }
    plan.pool.insert(plan.pool.end(), adj_loc.begin(), adj_loc.end());
    const int32_t meta_len = (int32_t)plan.pool.size() - off;
    while (plan.pool.size() % 4) plan.pool.push_back(0);   // keep every tile's block 16-B aligned

    // locality key = MEDIAN node BFS index the tile touches (for links: of their end nodes): node tiles and link tiles of
    // the same area sort next to each other, whatever shape the agglomeration gave them
    std::vector<int32_t> ks;
    for (int32_t r : own) {
      if (node_bfs) {
        for (int32_t p = inc.rowptr[r]; p < inc.rowptr[r + 1]; ++p) ks.push_back((*node_bfs)[inc.col[p]]);
      } else {
        ks.push_back(plan.bfs_index[r]);
      }
    }
    int32_t key = INT32_MAX;
    if (!ks.empty()) {
      std::nth_element(ks.begin(), ks.begin() + ks.size() / 2, ks.end());
      key = ks[ks.size() / 2];
    }
    plan.key.push_back(key);
    const int32_t h[TILE_HDR_INTS] = {(int32_t)own.size(), (int32_t)prim.size(), (int32_t)sec.size(), (int32_t)inc_loc.size(),
                                      (int32_t)adj_loc.size(), off, side, meta_len};
    plan.hdr.insert(plan.hdr.end(), h, h + TILE_HDR_INTS);
    plan.p_cap = std::max(plan.p_cap, ((int)prim.size() + 15) / 16 * 16);
    plan.q_cap = std::max(plan.q_cap, ((int)sec.size() + 15) / 16 * 16);
    plan.meta_cap = std::max(plan.meta_cap, (meta_len + 3) / 4 * 4);
    for (int32_t r : prim) ploc[r] = -1;
    for (int32_t q : sec) sloc[q] = -1;
  }
  return plan;
}

// ---- Tile blocks of k_fused_tile: fixed-width ("ELL") index lists -------------------------------------------------------
// The kernel walks a tile's index lists once per snapshot.  In CSR form every list costs a chain of dependent LDS reads
// (pointer -> entries -> rows); drainage networks have tiny degrees (a junction joins 2-4 conduits, a conduit has two
// ends), so the block a workgroup fetches stores them at fixed width instead -- one independent read per row:
//   hdr[8]              n_own, n_prim, n_sec, flags, n_ovf, n_adj (long-list tiles only), side, inc_width
//   prim[n_prim]        global primary rows (own rows first)
//   sec[n_sec]          global secondary rows
//   inc_loc[n_prim]     4 x u8: local secondary rows of a primary row's first four incident rows (0 = padding)
//   inc_w[4 n_prim]     their positions in the global NodeEdge value array, -1 = padding (the kernel puts the VALUES here)
//   adj[4 n_own]        16 x u8: local primary rows of an own row's first sixteen neighbours, 0xFF = none
//   flags & 1 (a primary row with more than four incident rows):    ovf_ptr[n_prim + 1], ovf_loc[n_ovf], ovf_w[n_ovf]
//   flags & 2 (an own row with more than sixteen neighbours):       adj_ptr[n_own + 1], adj_loc[n_adj]  (the whole lists)
// Every section starts on a multiple of four ints (16-byte LDS reads).  Local indices are bytes: tiles are limited to
// 255 primary / 256 secondary rows (ell_fits).
#ifdef __HIPCC__
#define UDS_HD __host__ __device__
#else
#define UDS_HD
#endif
constexpr int ELL_INC = 4, ELL_ADJ = 16;
constexpr int ELL_FLAG_INC_OVF = 1, ELL_FLAG_LONG_ROWS = 2;
UDS_HD inline int a4(int x) { return (x + 3) & ~3; }

struct EllOffsets {
  int prim, sec, inc_loc, inc_w, adj, ovf_ptr, ovf_loc, ovf_w, adj_ptr, adj_loc, len;
};
UDS_HD inline EllOffsets ell_offsets(int n_own, int n_prim, int n_sec, int flags, int n_ovf, int n_adj) {
  EllOffsets o{};
  int p = TILE_HDR_INTS;
  o.prim = p; p = a4(p + n_prim);
  o.sec = p; p = a4(p + n_sec);
  o.inc_loc = p; p = a4(p + n_prim);
  o.inc_w = p; p += ELL_INC * n_prim;
  o.adj = p; p += (ELL_ADJ / 4) * n_own;
  if (flags & ELL_FLAG_INC_OVF) {
    o.ovf_ptr = p; p = a4(p + n_prim + 1);
    o.ovf_loc = p; p = a4(p + n_ovf);
    o.ovf_w = p; p = a4(p + n_ovf);
  }
  if (flags & ELL_FLAG_LONG_ROWS) {
    o.adj_ptr = p; p = a4(p + n_own + 1);
    o.adj_loc = p; p = a4(p + n_adj);
  }
  o.len = p;
  return o;
}

// views of one tile inside a plan's pool (the CSR form build_side_plan writes)
struct TileView {
  int n_own, n_prim, n_sec, n_inc, n_adj, side;
  const int32_t *prim, *sec, *inc_ptr, *inc_loc, *inc_w, *adj_ptr, *adj_loc;
};
inline TileView tile_view(const int32_t *hd, const int32_t *pool) {
  TileView v{hd[0], hd[1], hd[2], hd[3], hd[4], hd[6], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  v.prim = pool + hd[5];
  v.sec = v.prim + v.n_prim;
  v.inc_ptr = v.sec + v.n_sec;
  v.inc_loc = v.inc_ptr + v.n_prim + 1;
  v.inc_w = v.inc_loc + v.n_inc;
  v.adj_ptr = v.inc_w + v.n_inc;
  v.adj_loc = v.adj_ptr + v.n_own + 1;
  return v;
}
inline void ell_shape(const TileView &v, int &flags, int &n_ovf, int &inc_width) {
  flags = 0; n_ovf = 0; inc_width = 0;
  for (int i = 0; i < v.n_prim; ++i) {
    const int d = v.inc_ptr[i + 1] - v.inc_ptr[i];
    inc_width = std::max(inc_width, std::min(d, ELL_INC));
    if (d > ELL_INC) { flags |= ELL_FLAG_INC_OVF; n_ovf += d - ELL_INC; }
  }
  for (int i = 0; i < v.n_own; ++i)
    if (v.adj_ptr[i + 1] - v.adj_ptr[i] > ELL_ADJ) flags |= ELL_FLAG_LONG_ROWS;
}
inline bool ell_fits(const int32_t *hd) { return hd[1] <= 255 && hd[2] <= 256; }
inline int ell_block_len(const int32_t *hd, const int32_t *pool) {
  const TileView v = tile_view(hd, pool);
  int flags, n_ovf, w;
  ell_shape(v, flags, n_ovf, w);
  return ell_offsets(v.n_own, v.n_prim, v.n_sec, flags, n_ovf, v.n_adj).len;
}
// largest block of a tile list (ints, a multiple of 4); -1 if some tile does not fit the byte-wide local indices
inline int ell_block_cap(const std::vector<int32_t> &hdr, const std::vector<int32_t> &pool, int n_tiles) {
  int cap = 4;
  for (int t = 0; t < n_tiles; ++t) {
    const int32_t *hd = hdr.data() + (size_t)t * TILE_HDR_INTS;
    if (!ell_fits(hd)) return -1;
    cap = std::max(cap, ell_block_len(hd, pool.data()));
  }
  return cap;
}
// blocks of a tile list at a fixed stride (>= ell_block_cap), zero padded
inline void build_ell_blocks(const std::vector<int32_t> &hdr, const std::vector<int32_t> &pool, int n_tiles, int stride,
                             std::vector<int32_t> &out) {
  out.assign(std::max<size_t>((size_t)stride * n_tiles, 4), 0);
  for (int t = 0; t < n_tiles; ++t) {
    const int32_t *hd = hdr.data() + (size_t)t * TILE_HDR_INTS;
    const TileView v = tile_view(hd, pool.data());
    int flags, n_ovf, w;
    ell_shape(v, flags, n_ovf, w);
    const EllOffsets o = ell_offsets(v.n_own, v.n_prim, v.n_sec, flags, n_ovf, v.n_adj);
    int32_t *b = out.data() + (size_t)stride * t;
    const int32_t h[TILE_HDR_INTS] = {v.n_own, v.n_prim, v.n_sec, flags, n_ovf, (flags & ELL_FLAG_LONG_ROWS) ? v.n_adj : 0, v.side, w};
    std::copy(h, h + TILE_HDR_INTS, b);
    std::copy(v.prim, v.prim + v.n_prim, b + o.prim);
    std::copy(v.sec, v.sec + v.n_sec, b + o.sec);
    int q = 0;
    if (flags & ELL_FLAG_INC_OVF) b[o.ovf_ptr] = 0;
    for (int i = 0; i < v.n_prim; ++i) {
      uint32_t locs = 0;
      for (int k = 0; k < ELL_INC; ++k) {
        const int p = v.inc_ptr[i] + k;
        const bool has = p < v.inc_ptr[i + 1];
        if (has) locs |= (uint32_t)v.inc_loc[p] << (8 * k);
        b[o.inc_w + ELL_INC * i + k] = has ? v.inc_w[p] : -1;
      }
      b[o.inc_loc + i] = (int32_t)locs;
      if (flags & ELL_FLAG_INC_OVF) {
        for (int p = v.inc_ptr[i] + ELL_INC; p < v.inc_ptr[i + 1]; ++p, ++q) {
          b[o.ovf_loc + q] = v.inc_loc[p];
          b[o.ovf_w + q] = v.inc_w[p];
        }
        b[o.ovf_ptr + i + 1] = q;
      }
    }
    for (int i = 0; i < v.n_own; ++i)
      for (int c = 0; c < ELL_ADJ / 4; ++c) {
        uint32_t word = 0;
        for (int k = 0; k < 4; ++k) {
          const int p = v.adj_ptr[i] + 4 * c + k;
          word |= (uint32_t)(p < v.adj_ptr[i + 1] ? v.adj_loc[p] : 0xFF) << (8 * k);
        }
        b[o.adj + (ELL_ADJ / 4) * i + c] = (int32_t)word;
      }
    if (flags & ELL_FLAG_LONG_ROWS) {
      std::copy(v.adj_ptr, v.adj_ptr + v.n_own + 1, b + o.adj_ptr);
      std::copy(v.adj_loc, v.adj_loc + v.n_adj, b + o.adj_loc);
    }
  }
}

// LDS bytes the fused kernel needs for a plan: meta + s_self/s_nbr + attention vectors + sec rows (h+4 stride) +
// hx rows + the DMA stage (raw secondary rows of width fs and raw primary rows of width fp of the NEXT snapshot,
// at least as large as the packed weights that pass through it once at workgroup start).
inline int64_t fused_lds_bytes(int p_cap, int q_cap, int meta_cap, int h, int d, int fp, int fs) {
  const int64_t weights = 16 * 64 * 2 * ((int64_t)(fs / 32) * (h / 16) + (int64_t)((fp + h) / 32) * (d / 16));
  const int64_t stage = std::max<int64_t>(4 * ((int64_t)q_cap * fs + (int64_t)p_cap * fp), weights);
#ifdef UDS_SMALL_IN_LDS
  const int64_t cold = (fp > 64 ? (d / 16) * 2 * 1024 : 0) + (int64_t)(fs / 32) * (h / 16) * 2 * 1024;
#else
  const int64_t cold = (fp > 64 ? (d / 16) * 2 * 1024 : 0) + (fs > 64 ? (h / 16) * 2 * 1024 : 0);   // LDS-resident third k-step
#endif
  return 4 * ((int64_t)meta_cap + 2 * p_cap + 2 * d + h + (int64_t)q_cap * (h + 4) + (int64_t)p_cap * d) + cold + stage;
}

// Both sides of a network merged into one tile list ordered by locality key, so that the node tile and
// the link tile that touch the same part of the network get neighbouring workgroup ids (same XCD L2).
struct NetworkPlan {
  SidePlan side[2];
  std::vector<int32_t> hdr;   // merged headers (pool_off rebased into `pool`)
  std::vector<int32_t> pool;
  int n_tiles = 0, p_cap = 0, q_cap = 0, meta_cap = 0, t_max[2] = {0, 0};
};

inline NetworkPlan build_network_plan(const HostCsr &adj, const HostCsr &eadj, const HostCsr &inc_n, const HostCsr &inc_e,
                                      int t_node, int t_link, int p_limit = 0, int q_limit = 0) {
  NetworkPlan np;
  np.t_max[0] = t_node;
  np.t_max[1] = t_link;
  np.side[0] = build_side_plan(adj, inc_n, t_node, 0, nullptr, p_limit, q_limit);
  np.side[1] = build_side_plan(eadj, inc_e, t_link, 1, &np.side[0].bfs_index, p_limit, q_limit);
  struct Ref { int32_t key, side, idx; };
  std::vector<Ref> refs;
  for (int s = 0; s < 2; ++s)
    for (int t = 0; t < np.side[s].n_tiles; ++t) refs.push_back({np.side[s].key[t], s, t});
  std::stable_sort(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.key < b.key; });
  const int32_t base1 = (int32_t)np.side[0].pool.size();
  np.pool = np.side[0].pool;
  np.pool.insert(np.pool.end(), np.side[1].pool.begin(), np.side[1].pool.end());
  for (const Ref &r : refs) {
    const int32_t *h = &np.side[r.side].hdr[(size_t)r.idx * TILE_HDR_INTS];
    for (int i = 0; i < TILE_HDR_INTS; ++i) np.hdr.push_back(i == 5 ? h[i] + (r.side ? base1 : 0) : h[i]);
  }
  np.n_tiles = (int)refs.size();
  np.p_cap = std::max(np.side[0].p_cap, np.side[1].p_cap);
  np.q_cap = std::max(np.side[0].q_cap, np.side[1].q_cap);
  np.meta_cap = std::max(np.side[0].meta_cap, np.side[1].meta_cap);
  return np;
}

}  // namespace uds
