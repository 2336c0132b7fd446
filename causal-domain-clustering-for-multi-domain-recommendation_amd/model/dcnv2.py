"""Deep & Cross Network v2 on the HIP hot path.  Mirror of the reference's model/dcnv2.py:9-70.

Only the constructor paths that work in the reference work here: use_low_rank_mixture=False and
model_structure="crossnet_only" raise AttributeError in the reference's __init__ (it registers
crossnet.u_list/v_list/c_list and self.dnn unconditionally, dcnv2.py:52-57) — golden g2_dcnv2_ctor_errors."""
import torch
from torch import nn

from .. import plan as P
from .layer import BaseModel, MultiLayerPerceptron, CrossNetV2, CrossNetMix, mlp_stack, _reg_filter


class DCNv2(BaseModel):
    def __init__(self, feature_dims, embed_dim, n_cross_layers, mlp_dims, dropout=0.2, model_structure="parallel",
                 use_low_rank_mixture=True, low_rank=32, num_experts=4,
                 l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.model_name = 'dcnv2'
        self.dropout_p = float(dropout)
        self.model_structure = model_structure
        assert self.model_structure in ["crossnet_only", "stacked", "parallel"], \
            "model_structure={} not supported!".format(self.model_structure)
        if use_low_rank_mixture:
            self.crossnet = CrossNetMix(self.embed_output_dim, n_cross_layers, low_rank=low_rank, num_experts=num_experts)
        else:
            self.crossnet = CrossNetV2(self.embed_output_dim, n_cross_layers)
        if self.model_structure == "stacked":
            self.dnn = MultiLayerPerceptron(self.embed_output_dim, mlp_dims, dropout, output_layer=False)
            final_dim = mlp_dims[-1]
        elif self.model_structure == "parallel":
            self.dnn = MultiLayerPerceptron(self.embed_output_dim, mlp_dims, dropout, output_layer=False)
            final_dim = mlp_dims[-1] + self.embed_output_dim
        else:
            final_dim = self.embed_output_dim
        self.mlp_out = mlp_dims[-1]
        self.dnn_linear = nn.Linear(final_dim, 1, bias=False)
        self.output_layer = nn.Sigmoid()
        # as in the reference, both of these raise AttributeError when the attribute does not exist
        self.add_regularization_weight(_reg_filter(self.dnn), l2=l2_reg_dnn)
        self.add_regularization_weight(self.dnn_linear.weight, l2=l2_reg_linear)
        for plist in [self.crossnet.u_list, self.crossnet.v_list, self.crossnet.c_list]:
            self.add_regularization_weight(plist, l2=l2_reg_cross)

    def describe(self, plan, emb):
        E = emb.out
        Ed = self.embed_output_dim
        if self.model_structure == "stacked":
            cross = self.crossnet.describe(plan, E)
            final = self.dnn.describe(plan, cross)
        else:                                   # parallel: cat[cross_out, dnn_out] built in place
            final = plan.new(Ed + self.mlp_out)
            self.crossnet.describe(plan, E, out=final.slice(0, Ed))
            mlp_stack(plan, [self.dnn], [E], last_outs=[final.slice(Ed, Ed + self.mlp_out)])
        wide = self.linear.describe(plan, E)
        out = plan.new(1)
        P.RowDot(plan, [{"x": final, "w": self.dnn_linear.weight, "b": None, "out": out}], addends=[wide], sigmoid=True)
        return [out], [], []

    def forward(self, x):
        return self.plan_holder(x.shape[0]).run(x.to(torch.int32)).squeeze(1)
