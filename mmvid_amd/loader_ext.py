"""Host-side mirror of mmvid_pytorch/loader_ext.py::VoxDataset (143-819): the Multimodal VoxCeleb frame-folder data set with its
attribute labels, segmentation masks and drawings (layout: mm_vox_celeb/README.md:12-42)

    <folder>/video/<key>/*.png      frames            <folder>/txt/<key>.txt     captions, one per line
    <folder>/label/<key>.txt        40 flags '0,1,..' <folder>/mask/<key>/*.png  <folder>/draw/style1/<key>/*.png

and the same constructor / return contract, so the reference's training loop (`text, frames, visuals = batch`, train.py:259-314)
and its `utils_train.py:17-44` dataset factory take it unchanged.  Same on-disk caches as the reference (`<folder>_local.pkl`: keys /
texts / videos / lengths with paths relative to the folder; `<folder>_attr_dict_vox2.pkl`: keys by person id and by attribute), so
a cache written by either side serves the other.  Host only: nothing here touches the device.

What a sample's control images and caption are is a TABLE here (`_RECIPES`): per `attr_mode` the image sources, whether their
order is drawn, and the caption rule -- the reference spells the same cases out as one branch each (loader_ext.py:470-787).

Parity: the image transforms restate torchvision's `Resize(size)` (shorter side, bilinear, antialias) + `CenterCrop` /
`RandomResizedCrop(scale=(resize_ratio, 1), ratio=(1, 1))`; torchvision and decord are absent from the build image, the reference
module cannot be imported here, so this file is checked against synthetic folders for structure and contract only (parity
unpinned for the random stream: the reference draws from the global `random`; pass `rng=` for a private source)."""
import os
import pickle
import random as _random
from pathlib import Path

import torch

from . import vox_text
from .data import IMG_EXT, _natural_key

ATTR, NAME, ATTR_VERB = vox_text.ATTR, vox_text.NAME, vox_text.ATTR_VERB
DRAW_STYLE = 'style1'  # loader_ext.py:468

# ---- what a sample's `visuals` and caption are, per attr_mode (loader_ext.py:470-787) -------------------------------------
# sources: 'own'   a frame of this video drawn with the clip (the `visual` of _get_video)      'mask' / 'draw'   of this video
#          'any'   a fresh random frame of this video                                           'pid:video' / 'pid:mask' / 'pid:draw'
#          of a random video of the same person (id#clip prefix); one such video is drawn per sample and shared by the sources
# caption: 'text' the file's caption | 'one' the fixed single-image sentence | 'pair' the two-image sentence naming both sources
#          in a drawn phrase order | 'motion' the image+video sentence;  dropout = probability of replacing the caption by "null"
_KIND = {'own': 'appearance', 'any': 'appearance', 'pid:video': 'appearance', 'mask': 'mask', 'pid:mask': 'mask', 'draw': 'draw',
         'pid:draw': 'draw'}
_RECIPES = {
    'text': dict(sources=['own'], caption='text'),
    'mask': dict(sources=['mask'], caption='one'),
    'draw': dict(sources=['draw'], caption='one'),
    'mask+text': dict(sources=['mask'], caption='text'),
    'mask+text_dropout': dict(sources=['mask'], caption='text', dropout=0.1, first_when_deterministic=True),
    'draw+text': dict(sources=['draw'], caption='text'),
    'draw+text_dropout': dict(sources=['draw'], caption='text', dropout=0.1),
    'image_same+draw': dict(sources=['own', 'draw'], caption='pair', shuffle=True),
    'image_same+mask': dict(sources=['own', 'mask'], caption='pair', shuffle=True),
    'image+draw': dict(sources=['pid:video', 'draw'], caption='pair', shuffle=True),
    'image+draw2': dict(sources=['pid:video', 'draw'], caption='pair'),
    'image+mask': dict(sources=['pid:video', 'mask'], caption='pair', shuffle=True),
    'image+mask2': dict(sources=['pid:video', 'mask'], caption='pair'),
    'draw+mask': dict(sources=['pid:draw', 'mask'], caption='pair', shuffle=True),
    'draw+mask2': dict(sources=['pid:draw', 'mask'], caption='pair'),
    'image+text_dropout': dict(sources=['pid_or_own'], caption='text', dropout=0.1, first_when_deterministic=True),
    'image+video33': dict(sources=['any'], caption='motion', motion=(3, 3)),
}
_ONE = 'A person in image one is talking'
_MOTION = 'A person with appearance in image one and motion in the following frames is talking.'
# cat2 (loader_ext.py:494-552): five fixed prompts; the first is about the indexed video itself, the others about a video that
# carries the attribute
_CAT2 = [('Male', None), ('Young', 'A person is youthful.'), ('Bald', 'A person has no hair.'), ('Eyeglasses', 'A person wears spectacles.'),
         ('Chubby', 'A person is plump.')]


def is_image_file(name):
    return name.lower().endswith(IMG_EXT)


def person_id(key):
    """'id11248#yDqlBD8m_b8#00004.txt#000.mp4' -> 'id11248#yDqlBD8m_b8' (loader_ext.py:262)."""
    return '#'.join(key.split('#')[:2])


def _open_rgb(path):
    import numpy as np
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert('RGB'), dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div_(255.0)


def _resize_short_side(x, size):
    """[..., H, W] -> shorter side = size, aspect kept (torchvision Resize(int))."""
    h, w = x.shape[-2:]
    if min(h, w) == size:
        return x
    nh, nw = (size, max(1, int(size * w / h))) if h <= w else (max(1, int(size * h / w)), size)
    lead = x.shape[:-3]
    y = torch.nn.functional.interpolate(x.reshape(-1, *x.shape[-3:]), size=(nh, nw), mode='bilinear', align_corners=False,
                                        antialias=True)
    return y.reshape(*lead, *y.shape[-3:])


def clip_transform(x, size, deterministic, resize_ratio, rng):
    """The image transform shared by the reference's clip data sets (loader.py:352-367, 716-731, 977-996; loader_ext.py:296-311):
    Resize(size) then CenterCrop(size) when deterministic, else RandomResizedCrop(size, scale=(resize_ratio, 1), ratio=(1, 1)) -- a
    square of that share of the image area, anywhere -- with ONE crop for the whole stack [..., 3, H, W]."""
    S = size
    x = _resize_short_side(x, S)
    h, w = x.shape[-2:]
    if deterministic:
        top, left = (h - S) // 2, (w - S) // 2
        return x[..., top:top + S, left:left + S].contiguous()
    side = int(round((rng.uniform(resize_ratio, 1.0) * h * w)**0.5))
    side = max(1, min(side, h, w))
    top, left = rng.randint(0, h - side), rng.randint(0, w - side)
    lead = x.shape[:-3]
    y = torch.nn.functional.interpolate(x[..., top:top + side, left:left + side].reshape(-1, *x.shape[-3:-2], side, side), size=(S, S),
                                        mode='bilinear', align_corners=False, antialias=True)
    return y.reshape(*lead, *y.shape[-3:])


class VoxDataset(torch.utils.data.Dataset):

    def __init__(self, folder, text_len=256, image_size=128, truncate_captions=False, resize_ratio=0.75, tokenizer=None, shuffle=False,
                 mode='video', frame_step=2, frame_num=8, deterministic=False, cache=None, return_vc=False, video_only=False, keys=None,
                 return_neg=False, attr_mode='mask+text', sample_label=False, cat1=(), args=None, rng=None):
        super().__init__()
        if mode not in ('video', '1frame'):
            raise NotImplementedError(f"VoxDataset mode {mode!r} (the reference builds only 'video' and '1frame', loader_ext.py:289-294)")
        self.root = str(folder)
        self.mode, self.shuffle, self.text_len, self.image_size = mode, shuffle, text_len, image_size
        self.truncate_captions, self.resize_ratio, self.tokenizer = truncate_captions, resize_ratio, tokenizer
        self.deterministic, self.sample_label, self.args = deterministic, sample_label, args
        self.frame_num, self.frame_step = frame_num, frame_step
        self.min_len = max(8, (frame_num - 1) * frame_step + 1)
        self.return_vc, self.return_neg, self.video_only = return_vc, return_neg, video_only
        self.attr_mode, self.cat1 = attr_mode, list(cat1)
        self.rng = rng or _random
        path = Path(folder)
        index_file = path.parent / (path.name + '_local.pkl') if cache is None else Path(cache)
        if index_file.exists():
            with open(index_file, 'rb') as fh:
                idx = pickle.load(fh)
            self.keys, self.texts, self.videos, self.lengths = idx['keys'], idx['texts'], idx['videos'], idx['lengths']
        else:
            self.keys, self.texts, self.videos, self.lengths = self._scan()
            with open(index_file, 'wb') as fh:
                pickle.dump(dict(root=self.root, keys=self.keys, texts=self.texts, videos=self.videos, lengths=self.lengths), fh)
        attr_file = path.parent / (path.name + '_attr_dict_vox2.pkl')
        if attr_file.exists():
            with open(attr_file, 'rb') as fh:
                attr = pickle.load(fh)
        else:
            attr = self._index_attributes()
            with open(attr_file, 'wb') as fh:
                pickle.dump(attr, fh)
        # videos too short for one clip are dropped (loader_ext.py:274-289); the attribute index follows
        keep = [k for k in self.keys if self.lengths[k] >= self.min_len]
        if keys is not None:
            wanted = set(keys)
            keep = [k for k in keep if k in wanted]
        kept = set(keep)
        self.keys = keep
        self.texts, self.videos = {k: self.texts[k] for k in keep}, {k: self.videos[k] for k in keep}
        self.lengths = {k: self.lengths[k] for k in keep}
        self.attr_dict = {kind: {a: [k for k in ks if k in kept] for a, ks in table.items()} for kind, table in attr.items()}
        assert len(self.keys) > 0, f'no usable videos under {self.root}/video'

    # ---- index --------------------------------------------------------------------------------------------------------------
    def _scan(self):
        vroot, troot = os.path.join(self.root, 'video'), os.path.join(self.root, 'txt')
        captions = set(os.listdir(troot))
        keys, texts, videos, lengths = [], {}, {}, {}
        for key in os.listdir(vroot):
            d = os.path.join(vroot, key)
            if not os.path.isdir(d) or key + '.txt' not in captions:
                continue
            frames = [os.path.join('video', key, f) for f in sorted(os.listdir(d), key=_natural_key) if is_image_file(f)]
            if frames:
                keys.append(key)
                texts[key], videos[key], lengths[key] = os.path.join('txt', key + '.txt'), frames, len(frames)
        assert keys, f'no videos with captions under {vroot}'
        return keys, texts, videos, lengths

    def _index_attributes(self):
        attr = {'pid': {}, 'attr': {}, 'cat1': {}}
        for k in self.keys:
            attr['pid'].setdefault(person_id(k), []).append(k)
            for j, flag in enumerate(self._get_label(k).split(',')):
                if flag == '1':
                    attr['cat1'].setdefault(j, []).append(k)
        return attr

    def _get_label(self, key):
        """The 40 comma-separated flags of label/<key>.txt (loader_ext.py:420-425)."""
        rel = self.texts[key]
        return Path(os.path.join(self.root, 'label' + rel[len('txt'):])).read_text().rstrip()

    # ---- images -------------------------------------------------------------------------------------------------------------
    def _transform(self, x):
        """[..., 3, H, W] in [0, 1] -> [..., 3, S, S]: one crop for the whole stack (loader_ext.py:296-311)."""
        return clip_transform(x, self.image_size, self.deterministic, self.resize_ratio, self.rng)

    def _frame(self, rel_path):
        return self._transform(_open_rgb(os.path.join(self.root, rel_path)))

    def _clip(self, key):
        """frame_num frames, frame_step apart, from a drawn start (loader_ext.py:313-324) -> ([T,3,S,S], start)."""
        n = self.lengths[key]
        start = 0 if self.deterministic else self.rng.randint(0, n - (self.frame_num - 1) * self.frame_step - 1)
        paths = self.videos[key][start:start + self.frame_num * self.frame_step:self.frame_step]
        return self._transform(torch.stack([_open_rgb(os.path.join(self.root, p)) for p in paths])), start

    def _get_video(self, index):
        key = self.keys[index]
        frames, start = self._clip(key)
        own = 0 if self.deterministic else self.rng.randint(0, self.lengths[key] - 1)
        return frames, key, self._frame(self.videos[key][own]), start

    def _get_video_by_key(self, key):
        return self._clip(key)

    def _get_1frame(self, index):
        """One frame from the middle three quarters of the video, and a second one as its control (loader_ext.py:349-366)."""
        key = self.keys[index]
        n = self.lengths[key]
        cut_r = int(n * 0.25 / 2)
        cut_l = int(n * 0.25) - cut_r
        a, b = self.rng.randint(cut_l, n - cut_r - 1), self.rng.randint(cut_l, n - cut_r - 1)
        return self._frame(self.videos[key][a]), key, self._frame(self.videos[key][b])

    def _folder_frame(self, *parts, first=False):
        d = os.path.join(self.root, *parts)
        names = os.listdir(d)
        return self._transform(_open_rgb(os.path.join(d, names[0] if first else self.rng.choice(names))))

    def _source(self, name, key, other, own, first):
        if name == 'own':
            return own
        if name == 'any':
            return self._folder_frame('video', key)
        if name == 'mask':
            return self._folder_frame('mask', key, first=first)
        if name == 'draw':
            return self._folder_frame('draw', DRAW_STYLE, key)
        if name == 'pid:video':
            return self._folder_frame('video', other)
        if name == 'pid:mask':
            return self._folder_frame('mask', other)
        if name == 'pid:draw':
            return self._folder_frame('draw', DRAW_STYLE, other)
        if name == 'pid_or_own':  # image+text_dropout: the same person's other clip half of the time (loader_ext.py:759-771)
            return self._folder_frame('video', other if self.rng.random() < 0.5 else key, first=first)
        raise KeyError(name)

    # ---- text ---------------------------------------------------------------------------------------------------------------
    def _tokenize_text(self, description):
        if self.tokenizer is None:
            return description
        return self.tokenizer.tokenize(description, self.text_len, truncate_text=self.truncate_captions).squeeze(0)

    def _captions(self, key):
        lines = Path(os.path.join(self.root, self.texts[key])).read_text().split('\n')
        return [t for t in lines if len(t) > 0]

    def _sample_negative_label(self, key):
        """A video whose attribute flags differ (loader_ext.py:427-434)."""
        label = self._get_label(key)
        while True:
            other = self.rng.choice(self.keys)
            if self._get_label(other) != label:
                return other

    # ---- protocol -----------------------------------------------------------------------------------------------------------
    def __len__(self):
        return len(self.keys)

    def random_sample(self):
        return self[self.rng.randint(0, len(self) - 1)]

    def sequential_sample(self, ind):
        return self[0] if ind >= len(self) - 1 else self[ind + 1]

    def skip_sample(self, ind):
        return self.random_sample() if self.shuffle else self.sequential_sample(ind)

    def _by_attribute(self, ind):
        """attr_mode 'cat1' / 'cat2': a stack of clips, one per attribute, each with a one-sentence caption about that attribute
        -> (clips [A,T,3,S,S], tokens [A,text_len])  (loader_ext.py:478-552)."""
        clips, texts = [], []
        if self.attr_mode == 'cat1':
            for yi in self.cat1:
                holders = self.attr_dict['cat1'][yi]
                clips.append(self._clip(holders[ind % len(holders)])[0])
                sentence = vox_text.generate_phrase((True, 1), (ATTR_VERB[ATTR[yi]], NAME[yi]), self.rng)  # pronoun form: "He ..."
                texts.append(self._tokenize_text('A person' + sentence[2:]))
        else:
            for attr, sentence in _CAT2:
                yi = ATTR.index(attr)
                if sentence is None:  # gender: the indexed video itself, named by its own flag
                    key = self.keys[ind]
                    male = self._get_label(key).split(',')[yi] == '1'
                    sentence = (('A boy.', 'A guy.') if male else ('A girl.', 'A lady.'))[ind % 2]
                else:
                    holders = self.attr_dict['cat1'][yi]
                    key = holders[ind % len(holders)]
                clips.append(self._clip(key)[0])
                texts.append(self._tokenize_text(sentence))
        return torch.stack(clips), (torch.stack(texts) if self.tokenizer is not None else texts)

    def __getitem__(self, ind):
        """-> (tokenized_text, frames, visuals [V,3,S,S]); with return_neg also (0, tokenized negative caption)."""
        if self.mode == 'video':
            frames, key, own, _start = self._get_video(ind)
        else:
            frames, key, own = self._get_1frame(ind)
        if self.video_only:
            return self._tokenize_text('dummy text'), frames, own
        if self.attr_mode in ('cat1', 'cat2'):
            return self._by_attribute(ind)
        captions = self._captions(key)
        if not captions:  # the reference's IndexError path: report and move on to another sample (loader_ext.py:788-791)
            print(f'An exception occurred trying to load file {os.path.join(self.root, self.texts[key])}.')
            print(f'Skipping index {ind}')
            return self.skip_sample(ind)
        description = captions[0] if self.deterministic else self.rng.choice(captions)
        recipe = _RECIPES.get(self.attr_mode, _RECIPES['text'])  # unknown modes fall back to the plain control frame
        first = bool(recipe.get('first_when_deterministic')) and self.deterministic
        other = None
        if any(s.startswith('pid') for s in recipe['sources']):
            other = self.rng.choice(self.attr_dict['pid'][person_id(key)])
        images = [self._source(s, key, other, own, first) for s in recipe['sources']]
        kinds = [_KIND.get(s, 'appearance') for s in recipe['sources']]
        if recipe.get('shuffle') and self.rng.random() >= 0.5:
            images, kinds = images[::-1], kinds[::-1]
        if recipe['caption'] == 'one':
            description = _ONE
        elif recipe['caption'] == 'pair':
            a, b = f'{kinds[0]} in image one', f'{kinds[1]} in image two'
            description = f'A person with {a} and {b} is talking' if self.rng.random() < 0.5 else f'A person with {b} and {a} is talking'
        elif recipe['caption'] == 'motion':
            num, step = recipe['motion']
            images += list(frames[:num * step:step])
            description = _MOTION
        if recipe.get('dropout') and self.rng.random() < recipe['dropout']:
            description = 'null'
        visuals = torch.stack(images)
        tokens = self._tokenize_text(description)
        if self.return_neg:
            negative = self._captions(self._sample_negative_label(key))
            return tokens, frames, visuals, 0, self._tokenize_text(self.rng.choice(negative))
        return tokens, frames, visuals
