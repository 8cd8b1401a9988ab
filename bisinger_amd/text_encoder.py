"""``TokenTextEncoder`` — the id mapping the model is built against (utils/text_encoder.py): ids 0/1/2 are
<pad>/<EOS>/<UNK>, the vocabulary follows.  Only what the hot path and the harness use (len, pad, encode)."""
PAD, EOS, UNK, SEG = '<pad>', '<EOS>', '<UNK>', '|'
RESERVED_TOKENS = [PAD, EOS, UNK]


class TokenTextEncoder:
    def __init__(self, vocab_filename, reverse=False, vocab_list=None, replace_oov=None):
        assert vocab_list is not None or vocab_filename, 'need a vocabulary'
        if vocab_list is None:
            with open(vocab_filename) as f:
                vocab_list = [ln.rstrip('\n') for ln in f]
        self._reverse = reverse
        self._replace_oov = replace_oov
        toks = RESERVED_TOKENS + [t for t in vocab_list if t not in RESERVED_TOKENS]
        self._id_to_token = dict(enumerate(toks))
        self._token_to_id = {t: i for i, t in self._id_to_token.items()}

    def __len__(self):
        return len(self._id_to_token)

    @property
    def vocab_size(self):
        return len(self)

    def pad(self):
        return 0

    def eos(self):
        return 1

    def unk(self):
        return 2

    def seg(self):
        return self._token_to_id.get(SEG)

    def encode(self, s):
        toks = s.strip().split() if isinstance(s, str) else list(s)
        if self._replace_oov is not None:
            toks = [t if t in self._token_to_id else self._replace_oov for t in toks]
        ids = [self._token_to_id[t] for t in toks]
        return ids[::-1] if self._reverse else ids

    def decode(self, ids, strip_eos=False, strip_padding=False):
        out = []
        for i in ids:
            i = int(i)
            if strip_padding and i == 0:
                continue
            if strip_eos and i == 1:
                break
            out.append(self._id_to_token.get(i, 'ID_%d' % i))
        return ' '.join(out)
