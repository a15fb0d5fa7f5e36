import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/cugraph_pyg_amd/sampler/sampler.py')
s = open(p).read()

# ---- homogeneous: with_replacement flag through neighbor_sample
s = s.replace('''def neighbor_sample(graph: CSRGraph, seeds: torch.Tensor, fanout: Sequence[int], random_state: int,
                    biased: bool = False, disjoint: bool = False, seed_time=None, temporal_comparison=None):''',
              '''def neighbor_sample(graph: CSRGraph, seeds: torch.Tensor, fanout: Sequence[int], random_state: int,
                    biased: bool = False, disjoint: bool = False, seed_time=None, temporal_comparison=None,
                    with_replacement: bool = False):''')
s = s.replace('''        else:
            off, nbr, lid, gid = wholegraph_ops.unweighted_sample_without_replacement(
                graph.row_ptr, graph.col, frontier, int(fan), hop_seed(random_state, k), True, True)
        new_nodes, mapping = graph_ops.append_unique(nodes, nbr, need_neighbor_raw_to_unique=True)''',
              '''        elif with_replacement:
            # `replace=True` (reference: forwarded to libcugraph, distributed_sampler.py:775-792): exactly `fan` picks
            # per vertex that has neighbours, repeats allowed
            off, nbr, lid, gid = wholegraph_ops.unweighted_sample_with_replacement(
                graph.row_ptr, graph.col, frontier, int(fan), hop_seed(random_state, k), True, True)
        else:
            off, nbr, lid, gid = wholegraph_ops.unweighted_sample_without_replacement(
                graph.row_ptr, graph.col, frontier, int(fan), hop_seed(random_state, k), True, True)
        new_nodes, mapping = graph_ops.append_unique(nodes, nbr, need_neighbor_raw_to_unique=True)''')
s = s.replace('''def _one_hop(graph: CSRGraph, frontier, fan, seed, biased):
    """(neighbours, row index in `frontier`, CSR slot) of one hop on one CSR; zero-weight edges dropped
    for biased sampling."""
    if biased:''', '''def _one_hop(graph: CSRGraph, frontier, fan, seed, biased, with_replacement=False):
    """(neighbours, row index in `frontier`, CSR slot) of one hop on one CSR; zero-weight edges dropped
    for biased sampling."""
    if with_replacement:
        off, nbr, lid, gid = wholegraph_ops.unweighted_sample_with_replacement(
            graph.row_ptr, graph.col, frontier, int(fan), seed, True, True)
        return nbr, lid, gid
    if biased:''')

# ---- heterogeneous: disjoint + with_replacement in hetero_neighbor_sample
s = s.replace('''def hetero_neighbor_sample(graphs, seed_type, seeds, fanout, random_state: int, biased: bool = False,
                           seed_time=None, temporal_comparison=None):''',
              '''def hetero_neighbor_sample(graphs, seed_type, seeds, fanout, random_state: int, biased: bool = False,
                           seed_time=None, temporal_comparison=None, disjoint: bool = False,
                           with_replacement: bool = False):''')
s = s.replace('''    Returns (node{type}, row{etype}, col{etype}, edge{etype}, num_sampled_nodes{type}[hops+1],
    num_sampled_edges{etype}[hops])."""''', '''    ``disjoint``: as in ``neighbor_sample`` — every seed grows its own tree across ALL node types, a vertex joins the tree
    of the first sampled edge that reaches it (edge types in sorted order inside a hop) and edges into another tree's
    vertex are dropped (trees are numbered over the seed types in sorted order).  ``with_replacement``: ``replace=True``.

    Returns (node{type}, row{etype}, col{etype}, edge{etype}, num_sampled_nodes{type}[hops+1],
    num_sampled_edges{etype}[hops])."""''')
s = s.replace('''    temporal = seed_time is not None
    if temporal:   # seed_time: tensor (single seed type) or {type: tensor}''', '''    tree = None
    if disjoint:   # tree id of every vertex of every type; seeds of the (sorted) seed types are numbered consecutively
        tree, base = {}, 0
        for t in ntypes:
            tree[t] = torch.arange(base, base + int(node[t].shape[0]), device=dev)
            base += int(node[t].shape[0])
    temporal = seed_time is not None
    if temporal:   # seed_time: tensor (single seed type) or {type: tensor}''')
s = s.replace('''                nbr, lid, gid = _one_hop(graphs[et], frontier, fan, hop_seed(random_state, h * len(etypes) + ti), biased)
            n_old = int(node[src_t].shape[0])
            new_nodes, mapping = graph_ops.append_unique(node[src_t], nbr, need_neighbor_raw_to_unique=True)
            if temporal and new_nodes.shape[0] > n_old:
                m = mapping.long()
                first = torch.full((new_nodes.shape[0] - n_old,), m.shape[0], dtype=torch.int64, device=dev)
                is_new = m >= n_old
                first.scatter_reduce_(0, m[is_new] - n_old, torch.arange(m.shape[0], device=dev)[is_new], reduce="amin")
                node_time[src_t] = torch.cat([node_time[src_t], graphs[et].time[gid[first]]])
            node[src_t] = new_nodes
            rows[et].append(mapping.long())
            cols[et].append(lid.long() + frontier_start[dst_t])
            edges[et].append(graphs[et].edge_id[gid])
            num_edges[et].append(int(nbr.shape[0]))''', '''                nbr, lid, gid = _one_hop(graphs[et], frontier, fan, hop_seed(random_state, h * len(etypes) + ti), biased,
                                         with_replacement)
            n_old = int(node[src_t].shape[0])
            new_nodes, mapping = graph_ops.append_unique(node[src_t], nbr, need_neighbor_raw_to_unique=True)
            m = mapping.long()
            src_row = lid.long() + frontier_start[dst_t]
            if (temporal or disjoint) and new_nodes.shape[0] > n_old:
                first = torch.full((new_nodes.shape[0] - n_old,), m.shape[0], dtype=torch.int64, device=dev)
                is_new = m >= n_old
                first.scatter_reduce_(0, m[is_new] - n_old, torch.arange(m.shape[0], device=dev)[is_new], reduce="amin")
                if temporal:
                    node_time[src_t] = torch.cat([node_time[src_t], graphs[et].time[gid[first]]])
                if disjoint:   # the FIRST edge that reaches a new vertex decides its tree
                    tree[src_t] = torch.cat([tree[src_t], tree[dst_t][src_row[first]]])
            node[src_t] = new_nodes
            if disjoint and nbr.shape[0] > 0:
                keep = tree[src_t][m] == tree[dst_t][src_row]
                m, src_row, gid, nbr = m[keep], src_row[keep], gid[keep], nbr[keep]
            rows[et].append(m)
            cols[et].append(src_row)
            edges[et].append(graphs[et].edge_id[gid])
            num_edges[et].append(int(nbr.shape[0]))''')

# ---- HeteroNeighborSampler: accept the flags, route to the one-batch path
s = s.replace('''        if with_replacement or disjoint:
            raise NotImplementedError("heterogeneous with_replacement / disjoint sampling are not implemented")
        if temporal and any(g.time is None for g in graphs.values()):''', '''        if with_replacement and (biased or temporal):
            raise NotImplementedError("sampling with replacement is uniform and non-temporal")
        self.with_replacement, self.disjoint = bool(with_replacement), bool(disjoint)
        if temporal and any(g.time is None for g in graphs.values()):''')
s = s.replace('''        return biased_ok and (not self.temporal) and (not getattr(self, "disjoint", False)) and all(
            g.col.dtype == torch.int64 for g in self.graphs.values())''', '''        return biased_ok and (not self.temporal) and (not self.disjoint) and (not self.with_replacement) and all(
            g.col.dtype == torch.int64 for g in self.graphs.values())''')
s = s.replace('''        fast = biased_ok and (not self.temporal) and seeds.is_cuda and all(
            g.col.dtype == torch.int64 for g in self.graphs.values())
        n_full = n // batch_size if fast else 0''', '''        fast = biased_ok and (not self.temporal) and (not self.disjoint) and (not self.with_replacement) and seeds.is_cuda and all(
            g.col.dtype == torch.int64 for g in self.graphs.values())
        n_full = n // batch_size if fast else 0''')
s = s.replace('''            yield bb, hetero_neighbor_sample(
                self.graphs, seed_type, seeds[start:start + batch_size], self.fanout, random_state + bb, self.biased,
                seed_time[start:start + batch_size] if self.temporal else None, self.temporal_comparison)''', '''            yield bb, hetero_neighbor_sample(
                self.graphs, seed_type, seeds[start:start + batch_size], self.fanout, random_state + bb, self.biased,
                seed_time[start:start + batch_size] if self.temporal else None, self.temporal_comparison,
                self.disjoint, self.with_replacement)''')

# ---- NeighborSampler: with_replacement
s = s.replace('''        if with_replacement:
            raise NotImplementedError("sampling with replacement is not implemented (kernels sample without)")
        if heterogeneous:''', '''        if with_replacement and (biased or temporal):
            raise NotImplementedError("sampling with replacement is uniform and non-temporal")
        self.with_replacement = bool(with_replacement)
        if heterogeneous:''')
open(p, 'w').write(s)
print("ok")
