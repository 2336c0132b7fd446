"""Causal Domain Clustering wrapper on the HIP hot path.

Mirror of the reference's model/cdc.py (CDC.__init__ 24-93, forward 95-111, get_matrix_metric 113-119, update_group
121-236, get_source_domain 238-293, update_p_weight 295-304, calc_metric_in_source_group 306-310,
get_center_domain_in_group 312-318, calc_domain_lambda_in_group 320-341, save/load_model_state 343-354,
get_regularization_loss 356-357, kmeans_group 359-362, calc_causal_matrix 364-396, save_draw_matrix 398-403).
The base model (MMoE / PLE / STAR / PEPNet / EPNet) runs on the HIP kernels; CDC itself only selects a tower output per row.  The
clustering arithmetic (O(n_domain^2) numpy/scipy/sklearn/torch work on 30x30 matrices, host side) lives in
`clustering.py` and is pinned against the reference's own outputs (tests/test_cdc_group.py)."""
import copy
import os
import re

import numpy as np
import torch
import torch.nn.functional as F

from .. import clustering
from .layer import BaseModel
from .mmoe import MMoE
from .pepnet import PEPNet
from .ple import PLE
from .star import STAR


class CDC(BaseModel):
    def __init__(self, feature_dims, embed_dim, n_tower, n_domain, base_model, expert_dims, tower_dims, domain_idx,
                 domain_cnt_weight=None, n_causal_mask=50, use_metric='loss', device='cpu', dropout=0.2, config=None,
                 savefig_folder='', l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5):
        super(BaseModel, self).__init__()            # like the reference: no embedding / linear of its own (cdc.py:29)
        self.model_name = 'cdc'
        self.base_model = base_model
        if base_model == 'mmoe':
            self.base_model_instance = MMoE(feature_dims, embed_dim, n_tower, config.mmoe_n_expert, expert_dims, tower_dims,
                                            dropout, config, l2_reg_embedding, l2_reg_linear, l2_reg_dnn, l2_reg_cross,
                                            model_name=self.model_name)
        elif base_model == 'ple':
            self.base_model_instance = PLE(feature_dims, embed_dim, n_tower, config.ple_n_expert_specific,
                                           config.ple_n_expert_shared, expert_dims, tower_dims, dropout, config,
                                           l2_reg_embedding, l2_reg_linear, l2_reg_dnn, l2_reg_cross, model_name=self.model_name)
        elif base_model == 'star':
            self.base_model_instance = STAR(feature_dims, embed_dim, n_tower, tower_dims, domain_idx, dropout, config,
                                            l2_reg_embedding, l2_reg_linear, l2_reg_dnn, l2_reg_cross, device)
        elif base_model in ('pepnet', 'epnet'):                       # cdc.py:43-50
            self.base_model_instance = PEPNet(feature_dims, embed_dim, n_tower, tower_dims, config.gate_hidden_dim, domain_idx,
                                              base_model == 'pepnet', dropout, config, l2_reg_embedding, l2_reg_linear,
                                              l2_reg_dnn, l2_reg_cross)
        else:
            raise ValueError('Unknown base model: ' + str(base_model))
        self.use_dcn = getattr(config, 'use_dcn', False)
        self.use_atten = getattr(config, 'use_atten', False)
        self.device = device
        self.config = config
        # cdc.py:58-60: the regrouping dumps its matrices under result/<dataset>/<folder>; created on first use here
        self.savefig_path = os.path.join('result', str(getattr(config, 'dataset_name', 'cdc')), savefig_folder)
        self.n_cluster = n_tower
        self.n_causal_mask = n_causal_mask
        self.n_domain = n_domain
        self.domain_idx = domain_idx
        self.domain_cnt_weight = None if domain_cnt_weight is None else torch.tensor(domain_cnt_weight, dtype=torch.float32, device=device)
        self.domain2group = torch.zeros(n_domain, dtype=torch.int64, device=device)
        self.domain2group_list = [0] * n_domain
        self.s_group2domain_list = [list(range(n_domain))]
        self.t_group2domain_list = [list(range(n_domain))]
        self.initial_s_group2domain_list = None
        self.call_update_group = 0
        self.p_weight = getattr(config, 'p_weight', None)
        self.p_weight_method = getattr(config, 'p_weight_method', None)
        self.matrix_A = torch.zeros((n_domain + 1, n_domain), dtype=torch.float32, device=device)
        self.matrix_B = torch.zeros((n_domain + self.n_cluster, n_domain), dtype=torch.float32, device=device)
        self.matrix_mask = torch.zeros((n_causal_mask, n_domain), dtype=torch.float32, device=device)
        self.matrix_causal = torch.zeros((n_causal_mask, n_domain), dtype=torch.float32, device=device)
        self.old_matrix_A, self.old_matrix_B, self.old_matrix_mask = None, None, None
        self.old_matrix_weight = getattr(config, 'old_matrix_weight', None)
        self.use_metric = use_metric
        if (self.use_metric == 'loss') ^ (getattr(config, 'affinity_func', None) == 'divide'):
            self.default_metric_value = 1e6
            self.is_max_metric_value_better = False
        else:
            self.default_metric_value = -1e6
            self.is_max_metric_value_better = True

    # ---- the forward surface (cdc.py:95-111) ---------------------------------------------------------
    def forward(self, x, mode='split', domain_i=None):
        y_cat = self.base_model_instance.forward(x)
        if mode == 'warmup':
            return torch.mean(y_cat, dim=1)
        if mode == 'split':
            if domain_i is None:
                groups = self.domain2group.to(x.device)[x[:, self.domain_idx].long()]
                return y_cat.gather(1, groups.unsqueeze(1))
            return y_cat[:, self.domain2group_list[domain_i]]

    def groups_of(self, x):
        """int64 [B]: the tower each row trains (what `split` mode gathers); feed it to TrainStep as `group`."""
        return self.domain2group.to(x.device)[x[:, self.domain_idx].long()]

    def get_matrix_metric(self, preds, targets):
        if self.use_metric == 'loss':
            return F.binary_cross_entropy(preds, targets).detach()
        from sklearn.metrics import roc_auc_score
        return roc_auc_score(targets.cpu().numpy(), preds.cpu().numpy())

    def save_model_state(self):
        pattern = re.compile('^(base_model_instance)')
        self.model_state = copy.deepcopy({k: v for k, v in self.state_dict().items() if pattern.match(k)})

    def load_model_state(self):
        self.load_state_dict(self.model_state, strict=False)

    def get_regularization_loss(self, device):
        return self.base_model_instance.get_regularization_loss(device)

    def set_precision(self, precision):
        self.base_model_instance.set_precision(precision)
        return self

    # ---- the regrouping (cdc.py:121-341, 359-403): arithmetic in clustering.py ------------------------------
    def update_group(self, mode='iterative'):
        return clustering.regroup(self, mode)

    def get_source_domain(self, t_group, group_idx):
        return clustering.select_source_domains(self, t_group, group_idx)

    def update_p_weight(self):
        clustering.decay_p_weight(self)

    def calc_metric_in_source_group(self, target_domain, s_group):
        return clustering.source_group_metric(self, target_domain, s_group)

    def get_center_domain_in_group(self, group, center_num=1):
        return clustering.group_centers(self.matrix_causal, group, center_num)

    def calc_domain_lambda_in_group(self, group, domain=None, mode='avg_dis'):
        if mode == 'avg_dis':
            return clustering.group_lambda(self.matrix_causal, group, domain, self.n_domain)

    @staticmethod
    def kmeans_group(matrix_causal, n_cluster):
        return clustering.kmeans_labels(matrix_causal, n_cluster)

    @staticmethod
    def calc_causal_matrix(X, alpha=None):
        return clustering.causal_kernel(X, alpha)

    def save_draw_matrix(self, matrix, name, is_illustration=False):
        """cdc.py:398-403 writes `<name> step-<k>.xlsx` through pandas/openpyxl; the same table goes out as .csv here
        (openpyxl is not a dependency of this package)."""
        if isinstance(matrix, torch.Tensor):
            matrix = matrix.cpu().numpy()
        os.makedirs(self.savefig_path, exist_ok=True)
        np.savetxt(os.path.join(self.savefig_path, f'{name} step-{self.call_update_group}.csv'), np.asarray(matrix), delimiter=',')
