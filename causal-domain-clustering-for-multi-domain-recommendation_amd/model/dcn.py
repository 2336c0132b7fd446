"""Deep & Cross Network (v1) on the HIP hot path.  Mirror of the reference's model/dcn.py:12-43."""
import torch
from torch import nn

from .. import plan as P
from .layer import BaseModel, MultiLayerPerceptron, CrossNetwork, mlp_stack, _reg_filter


class DCN(BaseModel):
    def __init__(self, feature_dims, embed_dim, n_cross_layers, mlp_dims, dropout=0.2,
                 l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.model_name = 'dcn'
        self.dropout_p = float(dropout)
        self.cn = CrossNetwork(self.embed_output_dim, n_cross_layers)
        self.mlp = MultiLayerPerceptron(self.embed_output_dim, mlp_dims, dropout, output_layer=False)
        self.mlp_linear = nn.Linear(self.embed_output_dim + mlp_dims[-1], 1, bias=False)
        self.output_layer = nn.Sigmoid()
        self.mlp_out = mlp_dims[-1]
        self.add_regularization_weight(_reg_filter(self.mlp), l2=l2_reg_dnn)
        self.add_regularization_weight(_reg_filter(self.cn), l2=l2_reg_cross)

    def describe(self, plan, emb):
        E = emb.out
        Ed = self.embed_output_dim
        # x_stack = cat[cross(e), mlp(e)] is never materialised by a copy: both producers write into its columns
        stack = plan.new(Ed + self.mlp_out)
        self.cn.describe(plan, E, out=stack.slice(0, Ed))
        mlp_stack(plan, [self.mlp], [E], last_outs=[stack.slice(Ed, Ed + self.mlp_out)])
        wide = self.linear.describe(plan, E)
        out = plan.new(1)
        P.RowDot(plan, [{"x": stack, "w": self.mlp_linear.weight, "b": None, "out": out}], addends=[wide], sigmoid=True)
        return [out], [], []

    def forward(self, x):
        return self.plan_holder(x.shape[0]).run(x.to(torch.int32)).squeeze(1)
