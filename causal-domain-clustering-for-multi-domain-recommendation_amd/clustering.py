"""The clustering arithmetic of Causal Domain Clustering — host-side control logic on n_domain x n_domain
matrices, no kernels.  Restated from the reference's model/cdc.py (file:line at each function) as free functions over an
explicit state object, so that `model/cdc.py`'s mirror methods stay thin; pinned against golden vectors produced by the
reference itself (tools/make_golden_cdc_group.py -> tests/golden/g10_*.npz, tests/test_cdc_group.py).

Decisions in here are arg-max/arg-min over float32 sums, so the arithmetic keeps the reference's operation order and its
torch dtypes wherever a value feeds a comparison."""
import copy

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------------
# dependence-contribution kernel (cdc.py:364-396, after Markham et al., "A Distance Covariance-based Kernel for
# Nonlinear Causal Clustering in Heterogeneous Populations")
# ---------------------------------------------------------------------------------------------------------
def causal_kernel(X, alpha=None):
    """X [samples, features] -> cosine-normalised kernel [samples, samples] (float64), entries clipped to <= 1."""
    if not isinstance(X, np.ndarray):
        X = X.cpu().numpy()
    n, f = X.shape
    thresh = np.eye(f)
    if alpha is not None:
        from scipy.stats import chi2
        crit = chi2(1).ppf(1 - alpha) / n                    # critical value of the pairwise test, off the diagonal
        thresh = np.where(np.eye(f) == 1, 0.0, crit)
    Z = np.empty((f, n, n))
    for j in range(f):
        col = X[:, j]
        dist = np.abs(col[:, None] - col[None, :])           # city-block distance of a single coordinate
        Z[j] = (dist - dist.mean(0) - dist.mean(1).reshape(-1, 1)) / dist.mean() + 1      # doubly centred, standardised
    flat = Z.reshape(f * n, n)
    mixed = np.tensordot(np.tensordot(Z, thresh, axes=([0], [0])), Z, axes=([2, 1], [0, 1]))
    gamma = (flat.T @ flat) ** 2 - 2 * mixed + np.linalg.norm(thresh)
    d = np.diag(gamma)
    kappa = gamma / np.sqrt(np.outer(d, d))
    kappa[kappa > 1] = 1
    return kappa


def kmeans_labels(matrix, n_cluster):
    """cdc.py:359-362: scikit-learn KMeans with its defaults (draws from numpy's global generator)."""
    from sklearn.cluster import KMeans
    return KMeans(n_clusters=n_cluster).fit(matrix).labels_


# ---------------------------------------------------------------------------------------------------------
# distances inside a group (cdc.py:320-341, 312-318, 306-310)
# ---------------------------------------------------------------------------------------------------------
def group_lambda(causal, group, domain=None, n_domain=None):
    """How close each `domain` is to `group` on the causal-distance matrix: (|group|-1) * related / unrelated / 2 in
    [0, 1], where related = summed distance group->domain and unrelated = the group's internal distance minus it."""
    inner = torch.sum(causal[np.ix_(group, group)])
    if domain is None:
        domain = list(range(n_domain if n_domain is not None else causal.shape[1]))
    related = torch.sum(causal[np.ix_(group, domain)], dim=0)
    vals = (len(group) - 1) * related / (inner - related) * 0.5
    return torch.clamp(vals, min=0, max=1)


def group_centers(causal, group, count=1):
    """the `count` members of `group` with the smallest lambda towards their own group"""
    count = min(count, len(group))
    _, order = torch.topk(group_lambda(causal, group, group), k=count, largest=False)
    return [group[i] for i in order]


def source_group_metric(st, target, s_group):
    lam = group_lambda(st.matrix_causal, s_group, [target])
    return torch.sum((1 - lam) * st.matrix_A[s_group, target] + lam * st.matrix_B[s_group, target])


# ---------------------------------------------------------------------------------------------------------
# source-domain selection for one target group (cdc.py:238-293)
# ---------------------------------------------------------------------------------------------------------
def select_source_domains(st, t_group, group_idx):
    """Greedy growth of the source set of target group `t_group`: start from its two centre domains, then keep adding the
    domain with the best weighted affinity J (+- the prior P from the initial clustering) while that affinity has the
    useful sign."""
    n = st.n_domain
    chosen = group_centers(st.matrix_causal, t_group, count=2)
    while len(chosen) < n:
        rows = []
        for d in range(n):
            if d in chosen:
                rows.append(torch.zeros(len(t_group), dtype=torch.float32, device=st.device))
            else:
                rows.append(group_lambda(st.matrix_causal, chosen + [d], t_group))
        lam = torch.stack(rows, dim=0)                                           # [n_domain, |t_group|]
        w = st.domain_cnt_weight[t_group]
        total = w.sum()
        if total != 0:
            w = w / total
        J = (((1 - lam) * st.matrix_A[:n, t_group] + lam * st.matrix_B[:n, t_group]) * w).sum(dim=1)
        if st.initial_s_group2domain_list is None:
            score = J
        else:
            prior = (1 - 2 * group_lambda(st.matrix_causal, st.initial_s_group2domain_list[group_idx], None, n)) * \
                torch.pow(st.domain_cnt_weight, 0.5)
            score = J + st.p_weight * prior if st.is_max_metric_value_better else J - st.p_weight * prior
        score[chosen] = st.default_metric_value
        if st.is_max_metric_value_better:
            best, who = torch.max(score, 0)
            useful = bool(best > 0)
        else:
            best, who = torch.min(score, 0)
            useful = bool(best < 0)
        if not useful:
            break
        chosen.append(who.item())
    return chosen


def decay_p_weight(st):
    """cdc.py:295-304"""
    if st.p_weight > 1e-10:
        if st.p_weight_method == 'linear_decay':
            st.p_weight = st.config.p_weight / st.call_update_group
        elif st.p_weight_method == 'quadratic_decay':
            st.p_weight = st.config.p_weight / (st.call_update_group ** 2)
        elif st.p_weight_method == 'exponential_decay':
            st.p_weight = st.p_weight * st.config.p_weight_exp_decay


# ---------------------------------------------------------------------------------------------------------
# one regrouping (cdc.py:121-236)
# ---------------------------------------------------------------------------------------------------------
def _best(st, t, dim=None):
    if dim is None:
        return torch.argmax(t) if st.is_max_metric_value_better else torch.argmin(t)
    return torch.argmax(t, dim=dim) if st.is_max_metric_value_better else torch.argmin(t, dim=dim)


def regroup(st, mode='iterative'):
    """`st` is the CDC module (or anything with its attributes).  Consumes matrix_A / matrix_B / matrix_mask as filled by
    the matrix-update loop (run.py:528-594), leaves the affinity-transformed matrices, the causal distance matrix and the
    new domain->cluster assignment on `st`, returns domain2group_list."""
    st.call_update_group += 1
    decay_p_weight(st)
    n, n_cluster = st.n_domain, st.n_cluster
    keep = st.old_matrix_weight
    if keep > 0 and st.old_matrix_A is not None:             # exponential smoothing over regroupings (A and B only)
        st.matrix_A = st.old_matrix_A * keep + st.matrix_A * (1 - keep)
        st.matrix_B = st.old_matrix_B * keep + st.matrix_B * (1 - keep)
    st.old_matrix_A, st.old_matrix_B, st.old_matrix_mask = (copy.deepcopy(st.matrix_A), copy.deepcopy(st.matrix_B),
                                                           copy.deepcopy(st.matrix_mask))
    warm = st.matrix_A[-1]                                   # metrics of the warmed-up model alone
    if st.config.affinity_func == 'minus':                   # smaller is better
        st.matrix_A[:-1] -= warm
        st.matrix_B[:n] = st.matrix_B[st.domain2group + n] - st.matrix_B[:n]
        st.matrix_mask = st.matrix_mask - warm
    elif st.config.affinity_func == 'divide':                # larger is better
        st.matrix_A[:-1] = 1 - st.matrix_A[:-1] / warm
        st.matrix_B[:n] = 1 - st.matrix_B[st.domain2group + n] / st.matrix_B[:n]
        st.matrix_mask = 1 - st.matrix_mask / warm
    else:
        raise ValueError('Unknown affinity_func: ' + st.config.affinity_func)
    st.matrix_causal = torch.tensor(np.arccos(st.calc_causal_matrix(st.matrix_mask.T)), dtype=torch.float32, device=st.device)
    for name in ('matrix_A', 'matrix_B', 'matrix_mask'):
        st.save_draw_matrix(getattr(st, name), name, is_illustration=True)
    st.save_draw_matrix(st.matrix_causal, 'causal_matrix', is_illustration=True)

    if max(st.domain2group_list) == 0:
        # first regrouping: clusters straight from the causal distances, then every cluster's source set
        labels = st.kmeans_group(st.matrix_causal.cpu().numpy(), n_cluster)
        st.domain2group_list = labels
        st.domain2group = torch.tensor(labels, dtype=torch.int64, device=st.device)
        members = [[] for _ in range(n_cluster)]
        for d, c in enumerate(labels):
            members[c].append(d)
        st.t_group2domain_list = members
        st.s_group2domain_list = [st.get_source_domain(members[c], group_idx=c) for c in range(n_cluster)]
        st.initial_s_group2domain_list = copy.deepcopy(st.s_group2domain_list)
        return st.domain2group_list

    previous = st.t_group2domain_list
    waiting = list(range(n))
    targets = [[] for _ in range(n_cluster)]
    sources = [[] for _ in range(n_cluster)]
    score = torch.empty(n, n_cluster)
    for c in range(n_cluster):                               # every cluster keeps the centre domain of its old members
        centre = st.get_center_domain_in_group(previous[c])[0]
        targets[c].append(centre)
        waiting.remove(centre)
        score[centre, :] = st.default_metric_value

    def rescore():
        for c in range(n_cluster):
            sources[c] = st.get_source_domain(targets[c], group_idx=c)
        for d in waiting:
            for c in range(n_cluster):
                score[d, c] = st.calc_metric_in_source_group(d, sources[c])

    if mode == 'iterative':
        # a waiting domain joins cluster c when it is c's best candidate AND c is its own best cluster; repeat
        moved = True
        while waiting and moved:
            moved = False
            rescore()
            candidate = _best(st, score, dim=0)
            for c in range(n_cluster):
                if _best(st, score[candidate[c], :]) == c:
                    moved = True
                    d = candidate[c].item()
                    targets[c].append(d)
                    waiting.remove(d)
                    score[candidate[c], :] = st.default_metric_value
        if waiting:
            raise ValueError('target domain_queue is not empty')
    elif mode == 'greedy':
        rescore()
        for d in waiting:
            targets[_best(st, score[d, :])].append(d)

    st.t_group2domain_list = targets
    assignment = np.array([0] * n)
    for c in range(n_cluster):
        st.s_group2domain_list[c] = st.get_source_domain(targets[c], group_idx=c)
        assignment[targets[c]] = c
    assignment = assignment.astype(int)
    st.domain2group = torch.tensor(assignment, dtype=torch.int64, device=st.device)
    st.domain2group_list = assignment.tolist()
    return st.domain2group_list
