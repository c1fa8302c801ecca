"""Two-stream encoder wrapper (encoders/dual_encoder.py:6-47 of the reference): stream_0 / stream_1 option
dicts override the shared args; optional weight tying; forward takes and returns 2-tuples."""
import copy
from argparse import Namespace

import torch.nn as nn


class DualEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        from . import str2encoder
        streams = []
        for over in (args.stream_0, args.stream_1):
            d = copy.deepcopy(vars(args))
            d.update(over)
            ns = Namespace(**d)
            streams.append(str2encoder[ns.encoder](ns))
        self.encoder_0, self.encoder_1 = streams
        if args.tie_weights:
            self.encoder_1 = self.encoder_0

    def forward(self, emb, seg):
        return self.get_encode_0(emb[0], seg[0]), self.get_encode_1(emb[1], seg[1])

    def get_encode_0(self, emb, seg):
        return self.encoder_0(emb, seg)

    def get_encode_1(self, emb, seg):
        return self.encoder_1(emb, seg)
