"""Host-side mirrors of the reference's one-FILE-per-video data sets (mmvid_pytorch/loader.py):

  TextImageStackDataset (852-1110)   `video/<id>.png` holds the frames side by side (or stacked vertically) as squares of the image's
                                     shorter side; `txt/<id>.txt` the captions; optional `label/<id>.txt`, `visual/<id>.png`
  TextMP4Dataset        (597-849)    `video/<id>.mp4` (any container the decoder opens) + `txt/<id>.txt`

Same constructors, the same `<folder>_local.pkl` index cache (keys / texts / videos / lengths, paths relative to the folder), the
same return tuples, so `utils_train.py:45-84` takes them unchanged.  Both are one class over a frame SOURCE (how many frames a file
holds, how a list of them is read); the reference writes the two classes out separately.  The mp4 source needs `decord`, which is
not in the build image: the class takes any `source=` with the same two methods (the tests use an in-memory one), and constructing
it without either raises an ImportError that says so.  Host only; parity unpinned (torchvision / decord absent: see loader_ext.py)."""
import os
import pickle
import random as _random
from pathlib import Path

import torch

from .data import IMG_EXT
from .loader_ext import clip_transform

VID_EXT = ('.mp4', '.avi', '.mov', '.mkv', '.webm', '.gif')


class ImageStackSource:
    """Frames stored in ONE image: n = longer side // shorter side squares, left to right or top to bottom (loader.py:128-141)."""
    extensions = IMG_EXT

    @staticmethod
    def _open(path):
        import numpy as np
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im.convert('RGB'), dtype=np.uint8)

    def count(self, path):
        h, w = self._open(path).shape[:2]
        return max(h, w) // min(h, w)

    def read(self, path, idxs=None):
        import numpy as np
        arr = self._open(path)
        h, w = arr.shape[:2]
        n, side = max(h, w) // min(h, w), min(h, w)
        frames = np.stack([arr[:, i * side:(i + 1) * side] if w > h else arr[i * side:(i + 1) * side] for i in range(n)])
        if idxs is not None:
            frames = frames[list(idxs)]
        return torch.from_numpy(frames.copy()).permute(0, 3, 1, 2).float().div_(255.0)


class DecordSource:
    """Frames of a video file through decord (loader.py:668-676, 747-757)."""
    extensions = VID_EXT

    def __init__(self):
        try:
            import decord
        except ImportError as e:  # pragma: no cover - decord is not part of the build image
            raise ImportError('TextMP4Dataset reads video files through decord (mmvid_pytorch/loader.py:15-17), which is not installed; '
                              'install it or pass source= (an object with count(path) and read(path, idxs))') from e
        decord.bridge.set_bridge('torch')
        self._decord = decord

    def count(self, path):
        return len(self._decord.VideoReader(path, num_threads=1))

    def read(self, path, idxs=None):
        reader = self._decord.VideoReader(path, num_threads=1)
        idxs = list(range(len(reader))) if idxs is None else list(idxs)
        return reader.get_batch(idxs).float().div_(255.0).permute(0, 3, 1, 2)


class _ClipFileDataset(torch.utils.data.Dataset):
    """Index + sampling shared by the two data sets; subclasses fix the source and the return tuple."""

    def __init__(self, folder, source, text_len, image_size, truncate_captions, resize_ratio, tokenizer, shuffle, mode, frame_step, frame_num,
                 deterministic, cache, keys, write_cache, rng):
        super().__init__()
        if mode not in ('video', '1frame'):
            raise NotImplementedError(f'mode {mode!r} (loader.py:705-710, 966-971 build only video / 1frame)')
        self.root, self.source = str(folder), source
        self.text_len, self.image_size, self.truncate_captions = text_len, image_size, truncate_captions
        self.resize_ratio, self.tokenizer, self.shuffle, self.mode = resize_ratio, tokenizer, shuffle, mode
        self.frame_step, self.frame_num, self.deterministic = frame_step, frame_num, deterministic
        self.min_len = max(8, (frame_num - 1) * frame_step + 1)
        self.rng = rng or _random
        path = Path(folder)
        self.has_label, self.has_visual = (path / 'label').exists(), (path / 'visual').exists()
        index_file = path.parent / (path.name + '_local.pkl') if cache is None else Path(cache)
        if index_file.exists():
            with open(index_file, 'rb') as fh:
                idx = pickle.load(fh)
        else:
            idx = self._scan()
            if write_cache:
                with open(index_file, 'wb') as fh:
                    pickle.dump(idx, fh)
        keep = [k for k in idx['keys'] if idx['lengths'][k] >= self.min_len]  # too short for one clip: dropped
        if keys is not None:
            wanted = set(keys)
            keep = [k for k in keep if k in wanted]
        self.keys = keep
        self.texts, self.videos = {k: idx['texts'][k] for k in keep}, {k: idx['videos'][k] for k in keep}
        self.lengths = {k: idx['lengths'][k] for k in keep}

    def _scan(self):
        captions = set(os.listdir(os.path.join(self.root, 'txt')))
        keys, texts, videos, lengths = [], {}, {}, {}
        for name in os.listdir(os.path.join(self.root, 'video')):
            vid = Path(name).stem
            if not name.lower().endswith(self.source.extensions) or vid + '.txt' not in captions:
                continue
            try:
                n = self.source.count(os.path.join(self.root, 'video', name))
            except Exception:  # an unreadable file is skipped, as in the reference
                continue
            keys.append(vid)
            texts[vid], videos[vid], lengths[vid] = os.path.join('txt', vid + '.txt'), os.path.join('video', name), n
        return dict(root=self.root, keys=keys, texts=texts, videos=videos, lengths=lengths)

    def _transform(self, x):
        return clip_transform(x, self.image_size, self.deterministic, self.resize_ratio, self.rng)

    def _visual(self, key, lo, hi):
        """The control frame: frame idx in [lo, hi] of `visual/<file>` when that folder exists, else of the video itself."""
        idx = self.rng.randint(lo, hi)
        rel = self.videos[key]
        path = os.path.join(self.root, 'visual', Path(rel).name) if self.has_visual else os.path.join(self.root, rel)
        return self._transform(self.source.read(path, [idx])[0])

    def _sample(self, index):
        key = self.keys[index]
        n = self.lengths[key]
        path = os.path.join(self.root, self.videos[key])
        if self.mode == 'video':
            start = self.rng.randint(0, n - (self.frame_num - 1) * self.frame_step - 1)  # inclusive, as random.randint
            idxs = range(start, start + self.frame_num * self.frame_step, self.frame_step)
            return self._transform(self.source.read(path, idxs)), key, (0, n - 1)
        cut_r = int(n * 0.25 / 2)  # one frame from the middle three quarters
        cut_l = int(n * 0.25) - cut_r
        return self._transform(self.source.read(path, [self.rng.randint(cut_l, n - cut_r - 1)])[0]), key, (cut_l, n - cut_r - 1)

    def _tokens(self, text):
        if self.tokenizer is None:
            return text
        return self.tokenizer.tokenize(text, self.text_len, truncate_text=self.truncate_captions).squeeze(0)

    def _caption(self, key, ind):
        lines = [t for t in Path(os.path.join(self.root, self.texts[key])).read_text().split('\n') if len(t) > 0]
        if not lines:
            print(f'An exception occurred trying to load file {os.path.join(self.root, self.texts[key])}.')
            print(f'Skipping index {ind}')
            return None
        return self.rng.choice(lines)

    def _label_text(self, key):
        rel = self.texts[key]
        return Path(os.path.join(self.root, 'label' + rel[len('txt'):])).read_text().rstrip()

    def __len__(self):
        return len(self.keys)

    def random_sample(self):
        return self[self.rng.randint(0, len(self) - 1)]

    def sequential_sample(self, ind):
        return self[0] if ind >= len(self) - 1 else self[ind + 1]

    def skip_sample(self, ind):
        return self.random_sample() if self.shuffle else self.sequential_sample(ind)


class TextImageStackDataset(_ClipFileDataset):

    def __init__(self, folder, text_len=256, image_size=128, truncate_captions=False, resize_ratio=0.75, tokenizer=None, shuffle=False,
                 mode='video', frame_step=2, frame_num=8, deterministic=False, image_only=False, cache=None, return_vc=False,
                 return_text=False, return_label=False, keys=None, no_cache=False, rng=None):
        super().__init__(folder, ImageStackSource(), text_len, image_size, truncate_captions, resize_ratio, tokenizer, shuffle, mode, frame_step,
                         frame_num, deterministic, cache, keys, not no_cache, rng)
        self.image_only, self.return_vc, self.return_text, self.return_label = image_only, return_vc, return_text, return_label

    def __getitem__(self, ind):
        """-> (tokens, frames [T,3,S,S]) [+ label int array | visual [3,S,S] | caption string], or (frames, 0) with image_only
        (loader.py:1072-1110)."""
        frames, key, span = self._sample(ind)
        if self.image_only:
            return frames, 0
        caption = self._caption(key, ind)
        if caption is None:
            return self.skip_sample(ind)
        tokens = self._tokens(caption)
        if self.return_label:
            import numpy as np
            return tokens, frames, np.array([int(v) for v in self._label_text(key).split(',')])
        if self.return_vc:
            return tokens, frames, self._visual(key, *span)
        if self.return_text:
            return tokens, frames, caption
        return tokens, frames


class TextMP4Dataset(_ClipFileDataset):

    def __init__(self, folder, text_len=256, image_size=128, truncate_captions=False, resize_ratio=0.75, tokenizer=None, shuffle=False,
                 mode='video', frame_step=2, frame_num=8, deterministic=False, image_only=False, cache=None, return_vc=False,
                 return_text=False, return_label=False, keys=None, video_only=False, source=None, rng=None):
        super().__init__(folder, source if source is not None else DecordSource(), text_len, image_size, truncate_captions, resize_ratio,
                         tokenizer, shuffle, mode, frame_step, frame_num, deterministic, cache, keys, True, rng)
        self.image_only, self.return_vc, self.return_text, self.return_label = image_only, return_vc, return_text, return_label
        self.video_only = video_only

    def __getitem__(self, ind):
        """-> (tokens, frames [T,3,S,S], visual [3,S,S]); return_label: the third item is the int of label/<id>.txt; video_only: the
        text is the 'dummy text' placeholder (loader.py:800-849)."""
        frames, key, span = self._sample(ind)
        visual = self._visual(key, *span)
        if self.video_only:
            return self._tokens('dummy text'), frames, visual
        caption = self._caption(key, ind)
        if caption is None:
            return self.skip_sample(ind)
        tokens = self._tokens(caption)
        if self.return_label:
            return tokens, frames, int(self._label_text(key))
        return tokens, frames, visual
