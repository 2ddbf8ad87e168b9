"""API alias for the reference's older hypernet.py (its own imports fail: hypernet.py:11 names
classes that exist only in the import-less later.py).  The constructor
signature HyperNet(embed_size, hidden_size, vocab_size, vocab, num_layers=1, type='gru', lr=1e-6)
(hypernet.py:27) maps onto the attention-GRU hypernet with feature_size = embed_size."""
from hypernet_attention import HyperNet as _AttentionHyperNet


class HyperNet(_AttentionHyperNet):
    def __init__(self, embed_size, hidden_size, vocab_size, vocab, num_layers=1, type='gru', lr=1e-6):
        if type != 'gru':
            raise NotImplementedError("only type='gru' is built (SURVEY.md 8f N3)")
        super().__init__(embed_size, embed_size, hidden_size, vocab_size, vocab, num_layers=num_layers, lr=lr)
