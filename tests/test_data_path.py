"""Host-side data path: tokenizer against goldens produced by the reference's tokenizer
(tools/make_golden_tokenizer.py), the frame-folder dataset on a synthetic folder, the gif / png writers."""
import json
import os

import pytest
import torch

from conftest import ROOT

VOCAB = os.environ.get('MMVID_BPE_VOCAB', '/root/reference/mmvid_pytorch/data/bpe_simple_vocab_16e6.txt')
needs_vocab = pytest.mark.skipif(not os.path.exists(VOCAB), reason='the CLIP merge table is reference data and is not shipped here')


@needs_vocab
def test_tokenizer_matches_reference_goldens():
    from mmvid_amd.data import SimpleTokenizer
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'tokenizer.json')))
    tok = SimpleTokenizer(VOCAB)
    assert tok.vocab_size == g['vocab_size'] == 49408 and len(tok.encoder) == 49408
    for r in g['rows']:
        ids = tok.encode(r['text'])
        assert ids == r['ids'], r['text']
        assert tok.decode(ids) == r['decoded'] and tok.decode(ids, remove_start_end=False) == r['decoded_keep']
    texts = [r['text'] for r in g['rows']]
    assert tok.tokenize(texts[:4], context_length=64).tolist() == g['tokenize_64']
    assert tok.tokenize([texts[8]], context_length=8, truncate_text=True).tolist() == g['truncate_8']
    with pytest.raises(RuntimeError, match='too long'):
        tok.tokenize([texts[8]], context_length=8)
    assert tok.decode(torch.tensor([49406, 320, 0, 0])) == tok.decode([320])  # start token and padding are dropped


def test_tokenizer_without_the_merge_table_fails_loudly(tmp_path, monkeypatch):
    from mmvid_amd.data import SimpleTokenizer
    monkeypatch.delenv('MMVID_BPE_VOCAB', raising=False)
    with pytest.raises(FileNotFoundError, match='merge table'):
        SimpleTokenizer(str(tmp_path / 'missing.txt'))


class _FakeTok:
    def tokenize(self, text, n, truncate_text=False):
        ids = [ord(c) % 251 + 1 for c in text][:n]
        out = torch.zeros(1, n, dtype=torch.long)
        out[0, :len(ids)] = torch.tensor(ids)
        return out


def _make_folder(root, lengths):
    import numpy as np
    from PIL import Image
    for key, n in lengths.items():
        os.makedirs(root / 'video' / key)
        for i in range(n):
            arr = np.full((40, 56, 3), (i * 9) % 256, np.uint8)
            arr[:, :, 1] = (len(key) * 31) % 256
            Image.fromarray(arr).save(root / 'video' / key / f'frame{i}.png')
    os.makedirs(root / 'txt')
    for key in list(lengths)[:-1]:  # the last video has no caption -> it is not part of the dataset
        (root / 'txt' / f'{key}.txt').write_text(f'caption of {key}\nsecond line ignored')


def test_text_video_dataset_layout_and_sampling(tmp_path):
    from mmvid_amd.data import TextVideoDataset
    _make_folder(tmp_path, {'clip_b': 20, 'clip_a': 12, 'short': 5, 'uncaptioned': 30})
    ds = TextVideoDataset(tmp_path, text_len=16, image_size=32, tokenizer=_FakeTok(), frame_step=2, frame_num=4, deterministic=True)
    assert ds.keys == ['clip_a', 'clip_b'] and ds.min_len == 8  # 'short' < max(8, 3*2+1) frames, 'uncaptioned' has no txt
    tokens, frames, visual = ds[1]  # the reference's contract (loader.py:500-562): `text, frames, visuals = batch`
    assert tokens.shape == (16, ) and tokens[0] == ord('c') % 251 + 1
    assert visual.shape == (3, 32, 32) and torch.equal(visual, frames[0])  # deterministic: frame 0 of the video
    assert frames.shape == (4, 3, 32, 32) and 0.0 <= float(frames.min()) and float(frames.max()) <= 1.0
    # natural order (frame10 after frame9) and a stride of two frames from frame 0: grey levels 0, 18, 36, 54
    assert [round(float(f[0, 0, 0]) * 255) for f in frames] == [0, 18, 36, 54]
    # the training transform: random start + one random-resized square crop for all frames, reproducible from a generator
    mk = lambda s: TextVideoDataset(tmp_path, text_len=16, image_size=32, tokenizer=_FakeTok(), frame_step=2, frame_num=4,
                                    generator=torch.Generator().manual_seed(s))
    a, b, c = mk(3)[1][1], mk(3)[1][1], mk(4)[1][1]
    assert torch.equal(a, b) and a.shape == (4, 3, 32, 32)
    starts = {round(float(mk(s)[1][1][0, 0, 0, 0]) * 255) // 9 for s in range(12)}
    assert len(starts) > 1 and max(starts) <= 20 - 6 - 1  # start in [0, len - span - 1], as random.randint's inclusive bound
    # video_only keeps the (text, frames, visual) order with the reference's 'dummy text' placeholder
    vo = TextVideoDataset(tmp_path, text_len=16, image_size=32, tokenizer=_FakeTok(), frame_num=4, video_only=True, deterministic=True)[0]
    assert len(vo) == 3 and vo[0].shape == (16, ) and vo[0][0] == ord('d') % 251 + 1 and vo[1].shape == (4, 3, 32, 32)
    # a random non-empty caption line when not deterministic
    (tmp_path / 'txt' / 'clip_b.txt').write_text('caption one\n\nzebra two\n')
    firsts = {int(mk(s)[1][0][0]) for s in range(16)}
    assert firsts == {ord('c') % 251 + 1, ord('z') % 251 + 1}


def test_save_image_tensor_gif_and_png(tmp_path):
    from PIL import Image

    from mmvid_amd.data import save_image_tensor
    video = torch.rand(1, 5, 3, 16, 24) * 1.4 - 0.2  # values outside [0, 1] are clamped
    name = save_image_tensor(video, tmp_path / 'sample')
    assert name == 'sample.gif'
    with Image.open(tmp_path / name) as im:
        assert im.n_frames == 5 and im.size == (24, 16)
    img = torch.linspace(0, 1, 3 * 8 * 8).view(3, 8, 8)
    assert save_image_tensor(img, tmp_path / 'still') == 'still.png'
    with Image.open(tmp_path / 'still.png') as im:
        import numpy as np
        px = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1)
    assert torch.equal(px, (img * 255).to(torch.uint8))  # truncation, as `(x * 255).type(torch.uint8)` in the reference
    with pytest.raises(NotImplementedError):
        save_image_tensor(video, tmp_path / 'x', video_format='mp4')


def test_reference_import_paths():
    """utils_train.py:25,187 import these names from `mmvid_pytorch.loader` / `.tokenizer`: the same module paths exist here."""
    from mmvid_amd import data
    from mmvid_amd.loader import TextVideoDataset
    from mmvid_amd.tokenizer import SimpleTokenizer
    assert TextVideoDataset is data.TextVideoDataset and SimpleTokenizer is data.SimpleTokenizer
