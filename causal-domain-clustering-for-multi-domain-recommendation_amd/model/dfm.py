"""DeepFM on the HIP hot path.  Mirror of the reference's model/dfm.py:9-35:
    y = sigmoid(linear(e) + fm(e) + mlp(e))      with mlp = [Linear, BatchNorm1d, ReLU, Dropout] x n + Linear(-> 1)
The wide term, the second-order FM term and the MLP's output layer meet in one row-dot launch (+ sigmoid)."""
import torch
from torch import nn

from .. import plan as P
from .layer import BaseModel, MultiLayerPerceptron, mlp_stack, _reg_filter


class DeepFM(BaseModel):
    def __init__(self, feature_dims, embed_dim, mlp_dims, dropout=0.2, l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.model_name = 'deepfm'
        self.dropout_p = float(dropout)
        self.mlp = MultiLayerPerceptron(self.embed_output_dim, mlp_dims, dropout, output_layer=True)
        self.add_regularization_weight(_reg_filter(self.mlp), l2=l2_reg_dnn)      # dfm.py:24-25: 'weight' in name, 'bn' not in name
        self.output_layer = nn.Sigmoid()

    def describe(self, plan, emb):
        E = emb.out
        wide = self.linear.describe(plan, E)
        fm = P.FMInteraction(plan, E, emb.F, emb.D).out
        outs, _ = mlp_stack(plan, [self.mlp], [E], final_addends=[wide, fm], final_sigmoid=True)
        return [outs[0]], [], []

    def forward(self, x):
        return self.plan_holder(x.shape[0]).run(x.to(torch.int32)).squeeze(1)
