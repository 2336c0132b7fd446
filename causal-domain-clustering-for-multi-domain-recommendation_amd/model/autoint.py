"""AutoInt on the HIP hot path.  Mirror of the reference's model/autoint.py:10-64:
    y = sigmoid(dnn_linear(cat[relu(interacting layers(e) + V_res(e)).flatten, dnn(e)]) + linear(e))
The interacting layers are the attention machinery of BaseModel's attention branch (model/layer.py:71-83): token embedding,
att_layer_num x MultiheadAttention over the field tokens, residual projection."""
import torch
from torch import nn

from .. import plan as P
from .layer import BaseModel, MultiLayerPerceptron, _reg_filter


class AutoInt(BaseModel):
    def __init__(self, feature_dims, embed_dim, atten_embed_dim=None, att_layer_num=3, att_head_num=2, att_res=True,
                 mlp_dims=(256, 128), dropout=0.2, l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.model_name = 'autoint'
        self.dropout_p = float(dropout)
        if len(mlp_dims) <= 0 and att_layer_num <= 0:
            raise ValueError("Either MLP hidden_layer or att_layer_num must > 0")
        if atten_embed_dim is None:
            atten_embed_dim = embed_dim
        self.atten_embedding = nn.Linear(embed_dim, atten_embed_dim)
        self.atten_output_dim = self.embedding.output_dim0 * atten_embed_dim
        self.att_res = att_res
        self.att_head_num = att_head_num
        self.dnn = MultiLayerPerceptron(self.embed_output_dim, mlp_dims, dropout, output_layer=False)
        self.self_attns = nn.ModuleList([nn.MultiheadAttention(atten_embed_dim, att_head_num, dropout=dropout)
                                         for _ in range(att_layer_num)])
        if self.att_res:
            self.V_res_embedding = nn.Linear(embed_dim, atten_embed_dim)
        self.mlp_out = mlp_dims[-1]
        self.dnn_linear = nn.Linear(self.mlp_out + self.atten_output_dim, 1, bias=False)
        self.output_layer = nn.Sigmoid()
        self.add_regularization_weight(_reg_filter(self.dnn), l2=l2_reg_dnn)

    def describe(self, plan, emb):
        from .layer import mlp_stack
        E = emb.out
        Fa = self.atten_output_dim
        stack = plan.new(Fa + self.mlp_out)                        # torch.cat((cross_term, dnn(e)), dim=1), never copied together
        self.describe_atten_features(plan, E, self.att_head_num, out=stack.slice(0, Fa))
        mlp_stack(plan, [self.dnn], [E], last_outs=[stack.slice(Fa, Fa + self.mlp_out)])
        wide = self.linear.describe(plan, E)
        out = plan.new(1)
        P.RowDot(plan, [{"x": stack, "w": self.dnn_linear.weight, "b": None, "out": out}], addends=[wide], sigmoid=True)
        return [out], [], []

    def forward(self, x):
        return self.plan_holder(x.shape[0]).run(x.to(torch.int32)).squeeze(1)
