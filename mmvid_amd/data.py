"""Host-side data path either side of the model: text -> token ids, a frame-folder video dataset
-> [T, 3, H, W] tensors in [0, 1], and decoded frames -> gif / png files.  Pure host code (no kernels): it exists so that a
training or sampling script written against the reference finds the same entry points here.

  SimpleTokenizer    mmvid_pytorch/tokenizer.py:61-171 (OpenAI CLIP byte-level BPE; 49,408 ids).  The merge table is the
                     reference's data file `mmvid_pytorch/data/bpe_simple_vocab_16e6.txt`; it is NOT shipped here -- pass its
                     path, or set MMVID_BPE_VOCAB.
  TextVideoDataset   mmvid_pytorch/loader.py:206-562, the layout `<root>/video/<key>/<frames>` + `<root>/txt/<key>.txt` and
                     the frame sampling (frame_num frames, frame_step apart; first caption line).  Built: mode='video', the
                     deterministic transform (resize + centre crop) and the random start / random-resized-crop of the training
                     transform with torch's generator; not built: pickle caches, negative sampling, 1frame mode.
  save_image_tensor  utils/utils_html.py:157-186 ([T,3,H,W] or [3,H,W] in [0,1] -> .gif / .png; mp4 needs torchvision.io).
"""
import html
import os
import re as _re

import torch

_SOT, _EOT = '<|startoftext|>', '<|endoftext|>'
IMG_EXT = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.tif', '.tiff', '.webp')


def _byte_symbols():
    """GPT-2's reversible byte -> printable-character table: the 188 printable latin-1 bytes stand for themselves, the other
    68 bytes are mapped, in increasing order, to the code points from 256 upwards."""
    keep = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
    table, extra = {b: chr(b) for b in keep}, 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table, keep + [b for b in range(256) if not (33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255)]


def _clean(text):
    """tokenizer.py:50-58: ftfy.fix_text (when ftfy is installed), two rounds of html.unescape, whitespace collapsed."""
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text)).strip()
    return _re.sub(r'\s+', ' ', text).strip()


class SimpleTokenizer:
    vocab_size = 49408

    def __init__(self, bpe_path=None):
        import regex
        bpe_path = bpe_path or os.environ.get('MMVID_BPE_VOCAB')
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError('SimpleTokenizer needs the CLIP merge table (mmvid_pytorch/data/bpe_simple_vocab_16e6.txt of the '
                                    'reference checkout): pass bpe_path= or set MMVID_BPE_VOCAB')
        table, order = _byte_symbols()
        self._byte_sym = table
        self._sym_byte = {c: b for b, c in table.items()}
        with open(bpe_path, encoding='utf8') as fh:
            lines = fh.read().split('\n')
        n_merges = self.vocab_size - 256 - 256 - 2  # ids: 256 byte symbols, 256 word-final ones, the merges, 2 specials
        pairs = [tuple(ln.split()) for ln in lines[1:1 + n_merges]]
        symbols = [table[b] for b in order]
        vocab = symbols + [s + '</w>' for s in symbols] + [a + b for a, b in pairs] + [_SOT, _EOT]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self._rank = {p: i for i, p in enumerate(pairs)}
        self._memo = {_SOT: [_SOT], _EOT: [_EOT]}
        self._split = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                    regex.IGNORECASE)

    def _merge(self, token):
        """Byte-pair merging of one pre-token (already in byte symbols): repeatedly fuse every occurrence of the adjacent pair
        with the lowest merge rank until no adjacent pair is in the table."""
        hit = self._memo.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + '</w>']
        while len(parts) > 1:
            best, best_rank = None, None
            for pair in zip(parts, parts[1:]):
                r = self._rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            fused, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    fused.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    fused.append(parts[i])
                    i += 1
            parts = fused
        self._memo[token] = parts
        return parts

    def encode(self, text):
        ids = []
        for piece in self._split.findall(_clean(text).lower()):
            sym = ''.join(self._byte_sym[b] for b in piece.encode('utf-8'))
            ids.extend(self.encoder[p] for p in self._merge(sym))
        return ids

    def decode(self, tokens, remove_start_end=True):
        if torch.is_tensor(tokens):
            tokens = tokens.tolist()
        if remove_start_end:  # the reference drops ids 49406, 40407 (sic) and the padding id 0 (tokenizer.py:149-152)
            tokens = [t for t in tokens if t not in (49406, 40407, 0)]
        text = ''.join(self.decoder[t] for t in tokens)
        return bytearray(self._sym_byte[c] for c in text).decode('utf-8', errors='replace').replace('</w>', ' ')

    def tokenize(self, texts, context_length=256, truncate_text=False):
        """-> int64 [len(texts), context_length], zero padded (tokenizer.py:157-171)."""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, text in enumerate(texts):
            ids = self.encode(text)
            if len(ids) > context_length:
                if not truncate_text:
                    raise RuntimeError(f'Input {text} is too long for context length {context_length}')
                ids = ids[:context_length]
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out


# ------------------------------------------------------------------------------------------------ dataset
def _natural_key(name):
    return [int(p) if p.isdigit() else p.lower() for p in _re.split(r'(\d+)', name)]


def _load_frame(path, size):
    """PIL image -> [3, size, size] float in [0, 1] (loader.py:414-417: Resize((size, size)) then to_tensor; bilinear)."""
    import numpy as np
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert('RGB').resize((size, size), Image.BILINEAR)
        arr = np.asarray(im, dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div_(255.0)


class TextVideoDataset(torch.utils.data.Dataset):
    """`folder/video/<key>/*.jpg` + `folder/txt/<key>.txt` -> (token ids [text_len], frames [frame_num, 3, S, S]).
    Videos shorter than max(8, (frame_num - 1) * frame_step + 1) frames are dropped (loader.py:252-262, 343-349)."""

    def __init__(self, folder, text_len=256, image_size=128, truncate_captions=False, resize_ratio=0.75, tokenizer=None,
                 frame_step=2, frame_num=8, deterministic=False, video_only=False, keys=None, generator=None, shuffle=False):
        super().__init__()
        self.root, self.text_len, self.image_size = str(folder), text_len, image_size
        self.truncate_captions, self.resize_ratio, self.tokenizer = truncate_captions, resize_ratio, tokenizer
        self.frame_step, self.frame_num, self.deterministic, self.video_only = frame_step, frame_num, deterministic, video_only
        self.generator = generator
        self.shuffle = shuffle  # loader.py:227, 247: decides what replaces a sample that cannot be loaded (skip_sample)
        self.min_len = max(8, (frame_num - 1) * frame_step + 1)
        vroot, troot = os.path.join(self.root, 'video'), os.path.join(self.root, 'txt')
        captions = set(os.listdir(troot)) if os.path.isdir(troot) else set()
        self.videos, self.texts = {}, {}
        for key in os.listdir(vroot):
            d = os.path.join(vroot, key)
            if not os.path.isdir(d) or (key + '.txt') not in captions:
                continue
            frames = sorted((f for f in os.listdir(d) if f.lower().endswith(IMG_EXT)), key=_natural_key)
            if len(frames) >= self.min_len:
                self.videos[key] = [os.path.join(d, f) for f in frames]
                self.texts[key] = os.path.join(troot, key + '.txt')
        if keys is not None:
            self.videos = {k: v for k, v in self.videos.items() if k in set(keys)}
        self.keys = sorted(self.videos)
        assert len(self.keys) > 0, f'no usable videos under {vroot}'

    def __len__(self):
        return len(self.keys)

    def _rand(self, n):
        return int(torch.randint(0, n, (1, ), generator=self.generator))

    def _frames(self, key):
        paths = self.videos[key]
        span = (self.frame_num - 1) * self.frame_step
        start = 0 if self.deterministic else self._rand(len(paths) - span)  # loader.py:396-398 (inclusive upper bound)
        x = torch.stack([_load_frame(paths[start + i * self.frame_step], self.image_size) for i in range(self.frame_num)])
        if not self.deterministic:  # RandomResizedCrop(scale=(resize_ratio, 1), ratio=(1, 1)): one square crop for all frames
            S = self.image_size
            area = float(torch.empty(1).uniform_(self.resize_ratio, 1.0, generator=self.generator)) * S * S
            side = max(1, min(S, int(round(area**0.5))))
            top, left = self._rand(S - side + 1), self._rand(S - side + 1)
            x = torch.nn.functional.interpolate(x[:, :, top:top + side, left:left + side], size=(S, S), mode='bilinear',
                                                align_corners=False, antialias=True)
        return x

    def _visual(self, key):
        """loader.py:418-422: one frame of the same video (frame 0 when deterministic, else uniformly random), through the same
        image transform -- the `visual` control the training loop passes on when --visual is set."""
        paths = self.videos[key]
        idx = 0 if self.deterministic else self._rand(len(paths))
        x = _load_frame(paths[idx], self.image_size).unsqueeze(0)
        if not self.deterministic:
            S = self.image_size
            area = float(torch.empty(1).uniform_(self.resize_ratio, 1.0, generator=self.generator)) * S * S
            side = max(1, min(S, int(round(area**0.5))))
            top, left = self._rand(S - side + 1), self._rand(S - side + 1)
            x = torch.nn.functional.interpolate(x[:, :, top:top + side, left:left + side], size=(S, S), mode='bilinear',
                                                align_corners=False, antialias=True)
        return x[0]

    def skip_sample(self, index):
        """loader.py:478-489: a random sample when the set is shuffled, else the next one (wrapping)."""
        if self.shuffle:
            return self[self._rand(len(self))]
        return self[0 if index >= len(self) - 1 else index + 1]

    def __getitem__(self, index):
        """-> (tokenized_text, frames [T,3,S,S], visual [3,S,S]) as loader.py:500-562 (`text, frames, visuals = batch`).
        video_only: the text is the reference's 'dummy text' placeholder."""
        key = self.keys[index]
        if self.video_only:
            caption = 'dummy text'
        else:  # (the caption is read first: a sample without one is replaced before any frame is decoded)
            with open(self.texts[key]) as fh:
                lines = [t for t in fh.read().split('\n') if len(t) > 0]  # loader.py:518-519: empty lines dropped
            if not lines:  # loader.py:533-536: a caption file without captions is skipped, not fatal
                print(f"An exception occurred trying to load file {self.texts[key]}.")
                print(f"Skipping index {index}")
                return self.skip_sample(index)
            caption = lines[0] if self.deterministic else lines[self._rand(len(lines))]  # loader.py:521-524
        frames = self._frames(key)
        visual = self._visual(key)
        tokens = self.tokenizer.tokenize(caption, self.text_len, truncate_text=self.truncate_captions).squeeze(0)
        return tokens, frames, visual


# ------------------------------------------------------------------------------------------------ output side
def _box(kind, *payload):
    body = b''.join(payload)
    return (8 + len(body)).to_bytes(4, 'big') + kind + body


def _u(value, nbytes=4):
    return int(value).to_bytes(nbytes, 'big')


_UNITY = b''.join(_u(v) for v in (0x00010000, 0, 0, 0, 0x00010000, 0, 0, 0, 0x40000000))  # the identity transformation matrix


def write_mjpeg_mp4(path, frames_u8, fps=4, quality=95):
    """frames_u8 [T, H, W, 3] uint8 -> an ISO base-media (.mp4) file whose video track is Motion-JPEG: every frame a baseline JPEG
    (PIL), all samples in one chunk of `mdat`, constant frame duration.  The reference writes H.264 through
    torchvision.io.write_video (utils_html.py:178-184: PyAV, absent from this image); Motion-JPEG needs no codec library, is
    intra-only (every frame a key frame) and is read by ffmpeg / VLC / QuickTime.  Returns the list of (offset, size) of the frames."""
    import io

    from PIL import Image
    T, H, W, _ = frames_u8.shape
    jpegs = []
    for f in frames_u8:
        buf = io.BytesIO()
        Image.fromarray(f.numpy() if hasattr(f, 'numpy') else f).save(buf, format='JPEG', quality=quality, subsampling=0)
        jpegs.append(buf.getvalue())
    scale = 1000
    delta = max(1, round(scale / fps))
    duration = delta * T
    ftyp = _box(b'ftyp', b'isom', _u(0x200), b'isom', b'iso2', b'mp41')
    first = len(ftyp) + 8  # the frames follow the mdat header, which follows ftyp
    mdat = _box(b'mdat', *jpegs)
    mvhd = _box(b'mvhd', _u(0), _u(0), _u(0), _u(scale), _u(duration), _u(0x00010000), _u(0x0100, 2), bytes(10), _UNITY, bytes(24), _u(2))
    tkhd = _box(b'tkhd', _u(3), _u(0), _u(0), _u(1), _u(0), _u(duration), bytes(8), _u(0, 2), _u(0, 2), _u(0, 2), _u(0, 2), _UNITY,
                _u(W << 16), _u(H << 16))
    mdhd = _box(b'mdhd', _u(0), _u(0), _u(0), _u(scale), _u(duration), _u(0x55C4, 2), _u(0, 2))
    hdlr = _box(b'hdlr', _u(0), _u(0), b'vide', bytes(12), b'VideoHandler\x00')
    name = b'Motion JPEG'
    entry = _box(b'jpeg', bytes(6), _u(1, 2), bytes(16), _u(W, 2), _u(H, 2), _u(0x00480000), _u(0x00480000), _u(0), _u(1, 2),
                 bytes([len(name)]) + name + bytes(31 - len(name)), _u(0x0018, 2), _u(0xFFFF, 2))
    stsd = _box(b'stsd', _u(0), _u(1), entry)
    stts = _box(b'stts', _u(0), _u(1), _u(T), _u(delta))
    stsc = _box(b'stsc', _u(0), _u(1), _u(1), _u(T), _u(1))
    stsz = _box(b'stsz', _u(0), _u(0), _u(T), *[_u(len(j)) for j in jpegs])
    stco = _box(b'stco', _u(0), _u(1), _u(first))
    stbl = _box(b'stbl', stsd, stts, stsc, stsz, stco)
    dinf = _box(b'dinf', _box(b'dref', _u(0), _u(1), _box(b'url ', _u(1))))
    minf = _box(b'minf', _box(b'vmhd', _u(1), bytes(8)), dinf, stbl)
    moov = _box(b'moov', mvhd, _box(b'trak', tkhd, _box(b'mdia', mdhd, hdlr, minf)))
    with open(path, 'wb') as fh:
        fh.write(ftyp + mdat + moov)
    spans, off = [], first
    for j in jpegs:
        spans.append((off, len(j)))
        off += len(j)
    return spans


@torch.no_grad()
def save_image_tensor(tensor, path, video_format='gif', fps=4):
    """utils/utils_html.py:157-186.  [3,H,W] / [1,3,H,W] -> `<path>.png`; [T,3,H,W] / [1,T,3,H,W] -> `<path>.gif` or, with
    video_format='mp4', `<path>.mp4` (Motion-JPEG, `write_mjpeg_mp4`; the reference: H.264 at the same 4 frames per second).
    Values are clamped to [0, 1] and quantised by truncation (`* 255` then uint8), as the reference does.  Returns the file name."""
    from PIL import Image
    t = tensor.squeeze(0) if tensor.dim() in (4, 5) and tensor.shape[0] == 1 else tensor
    u8 = (t.detach().float().cpu().clamp(0, 1) * 255).to(torch.uint8)
    if u8.dim() == 3:
        out = str(path) + '.png'
        Image.fromarray(u8.permute(1, 2, 0).numpy()).save(out)
    elif u8.dim() == 4:
        if video_format == 'gif':
            out = str(path) + '.gif'
            frames = [Image.fromarray(f.permute(1, 2, 0).numpy()) for f in u8]
            frames[0].save(out, save_all=True, append_images=frames[1:], duration=int(1000 / fps), loop=0)
        else:
            out = str(path) + '.mp4'
            write_mjpeg_mp4(out, u8.permute(0, 2, 3, 1).contiguous(), fps=fps)
    else:
        raise RuntimeError(f'save_image_tensor: unsupported shape {tuple(tensor.shape)}')
    return os.path.basename(out)
