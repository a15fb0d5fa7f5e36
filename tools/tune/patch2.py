import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_sage_mfma.hip')
s = open(p).read()

# ids: self row index as int32 (local rows of the batch; the API hands int64, rows < 2^31 by construction of x_rows)
s = s.replace('''template <int IT>
struct ids_t {
  int lcol[IT];
  int64_t lself[IT];
};''', '''template <int IT>
struct ids_t {
  int deg[IT];     // e - s of the bounds (the bounds die when the ids are requested)
  int lcol[IT];
  int lself[IT];   // self row (a local row of x / src_ids: < 2^31)
};''')
old = s[s.index('  // stage B: this lane\'s neighbour id of every row'):s.index('  // request the kNb neighbour rows + the self row of row `it`')]
new = '''  // stage B: this lane's neighbour id of every row + the self row ids (requested mid-tile for the next tile); unconditional
  __device__ __forceinline__ void load_ids(int64_t tile, const bounds_t<IT>& b, ids_t<IT>& v) const
  {
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int64_t row = row_of(tile, it);
      v.deg[it]         = row < a.n_rows ? b.e[it] - b.s[it] : -1;
      const int* pc     = (sub < b.e[it] - b.s[it]) ? a.col + b.s[it] + sub : a.row_ptr;  // row_ptr[0] == 0: a valid row
      v.lcol[it]        = *pc;
      v.lself[it]       = (int)a.self_rows[row < a.n_rows ? row : a.n_rows - 1];
    }
  }
  // stage C: byte offsets (with the id indirection of the fused-fetch variant: one more dependent load)
  __device__ __forceinline__ void finish(const ids_t<IT>& v, meta_t<IT, off_t>& m) const
  {
    const IdT* src_ids = static_cast<const IdT*>(a.src_ids);
#pragma unroll
    for (int it = 0; it < IT; it++) {
      m.d[it]    = v.deg[it];
      m.src[it]  = (off_t)(table_row<IdT>(src_ids, (int64_t)v.lcol[it]) * a.ldx * 4);
      m.self[it] = (off_t)(table_row<IdT>(src_ids, (int64_t)v.lself[it]) * a.ldx * 4);
    }
  }
'''
s = s.replace(old, new)

old = s[s.index('    bounds_t<IT> b_next, b_next2;   // bounds of tile n+1, n+2'):s.index('      lds_barrier();\n    }\n  } else {')]
new = '''    // metadata of tile n+1 is fetched WHILE tile n is summed, in three dependent stages spread over the tile so that no
    // stage ever waits: CSR bounds at the start of the tile, neighbour / self ids (they need the bounds) half-way, byte
    // offsets (they need the ids) at the end.  Bounds and ids are never live together: ~48 metadata registers instead of 80,
    // which is what lets the row ring be kDepth = 3 deep.
    bounds_t<IT> b_next;
    ids_t<IT> i_next;
    meta_t<IT, off_t> cur;
    f32x4 buf[kDepth][kNb + 1];
    {
      p.load_bounds(tile_of(0), b_next);
      p.load_ids(tile_of(0), b_next, i_next);
      p.finish(i_next, cur);
    }
#pragma unroll
    for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
    constexpr int kHalf = IT / 2;
    for (int64_t n = 0; n <= mine; n++) {
      if (n < mine && !(a.debug & 2)) {
        uint32_t* tile_lds = lds + (n & 1) * tile_dw;
        p.load_bounds(tile_of(n + 1), b_next);
#pragma unroll
        for (int it = 0; it < IT; it++) {
          if (it == kHalf) p.load_ids(tile_of(n + 1), b_next, i_next);
          if (it + kDepth - 1 < IT) p.issue(cur, it + kDepth - 1, buf[(it + kDepth - 1) % kDepth]);
          p.reduce_store(cur, it, buf[it % kDepth], tile_lds);
        }
        p.long_rows(tile_of(n), cur, tile_lds);
        // offsets of tile n+1, and its first rows in flight BEFORE the barrier
        p.finish(i_next, cur);
        if (n + 1 < mine) {
#pragma unroll
          for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
        }
      }
'''
s = s.replace(old, new)
s = s.replace("#define WG_MFMA_DEPTH 2   // destination rows in flight per producer lane group", "#define WG_MFMA_DEPTH 3   // destination rows in flight per producer lane group")
open(p, 'w').write(s)
p = os.path.join(ROOT, 'tools/tune/build.sh')
s = open(p).read()
s = s.replace('for v in "4 2 100" "4 2 0"; do', 'for v in "4 2 100" "4 3 100" "4 4 100"; do')
open(p, 'w').write(s)
print("ok")
