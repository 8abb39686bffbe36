"""Golden vectors for mmvid_amd.data.SimpleTokenizer: token ids and decoded strings produced by the REFERENCE's tokenizer
(mmvid_pytorch/tokenizer.py) for a fixed list of captions.  Build container only (imports /root/reference; `ftfy`, which is
not installed here, is replaced by the identity -- every caption below is already well-formed text, where ftfy.fix_text is the
identity).  Writes tests/golden/tokenizer.json."""
import json
import os
import sys
import types

REF = '/root/reference'
sys.modules['ftfy'] = types.SimpleNamespace(fix_text=lambda t: t)
sys.path.insert(0, os.path.join(REF, 'mmvid_pytorch'))
import tokenizer as ref_tok  # noqa: E402

CAPTIONS = [
    'a person is talking',
    'She has blond hair, arched eyebrows and is wearing lipstick.',
    "The man's beard isn't grey; he's 45 years old, I'd say.",
    'An   object   with\ttabs and\nnewlines  ',
    'Unicode: café naïve Zürich — “quoted” text… and 日本語のテキスト',
    'numbers 1234567890 and symbols !@#$%^&*()_+-=[]{}|;:,.<>/?',
    'HTML &amp;amp; entities &lt;b&gt; twice &amp;quot;escaped&amp;quot;',
    'supercalifragilisticexpialidocious antidisestablishmentarianism',
    'the red cube moves to the left of the blue sphere while the green cylinder rotates',
    '',
    'emoji 🙂🚀 and mixed CASE WoRdS',
    "it's we're they've i'm you'll he'd",
]
tok = ref_tok.SimpleTokenizer()
rows = []
for c in CAPTIONS:
    ids = tok.encode(c)
    rows.append({'text': c, 'ids': ids, 'decoded': tok.decode(ids), 'decoded_keep': tok.decode(ids, remove_start_end=False)})
padded = tok.tokenize(CAPTIONS[:4], context_length=64).tolist()
trunc = tok.tokenize([CAPTIONS[8]], context_length=8, truncate_text=True).tolist()
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'tokenizer.json')
with open(out, 'w') as fh:
    json.dump({'rows': rows, 'tokenize_64': padded, 'truncate_8': trunc, 'vocab_size': tok.vocab_size}, fh)
print('wrote', out, sum(len(r['ids']) for r in rows), 'ids')
