"""Attribute sentences of the Multimodal VoxCeleb data: the probabilistic context-free grammar of
mm_vox_celeb/pcfg.py (the reference's `make_text.py` writes `txt/<key>.txt` with it, and `VoxDataset(attr_mode='cat1')` builds
single-attribute captions with `generate_phrase`).  Host-side text generation only; nothing here touches the device.

Grammar (pcfg.py:119-181):   S  -> NP VP '.'
                             NP -> pronoun                      with probability p_pronoun
                                 | Det Noun                     Det in {a, this}; Noun gendered with probability 0.75, else 'person'
                             VP -> 'is' A | 'has' A | ('wears' | 'is wearing') A
An attribute list (pcfg.py:80-117) is first negated where the label is a 'No_*' one, shuffled, split by verb, and merged into
groups of one to three attributes ("a, b and c") -- group sizes by the probabilities of `merge_and_pop` (pcfg.py:104-117).

Parity: the reference draws from BOTH `random` and `numpy.random`; here one `random.Random`-like source is used, so sentences agree
in distribution and form, not draw for draw (parity unpinned: tests check the grammar, not a recorded stream)."""
import random as _random

# the 40 CelebA attribute names in label-file order (pcfg.py:11-22 == loader_ext.py:21-32) and the verb each one takes (pcfg.py:33-74)
_VERBS = {
    'has': ('5_o_Clock_Shadow Arched_Eyebrows Bags_Under_Eyes Bangs Big_Lips Big_Nose Black_Hair Blond_Hair Brown_Hair Bushy_Eyebrows '
            'Double_Chin Gray_Hair Heavy_Makeup High_Cheekbones Mustache Narrow_Eyes No_Beard Oval_Face Pale_Skin Pointy_Nose '
            'Receding_Hairline Rosy_Cheeks Sideburns Straight_Hair Wavy_Hair').split(),
    'is': 'Attractive Bald Blurry Chubby Male Smiling Young'.split(),
    'wear': 'Eyeglasses Goatee Wearing_Earrings Wearing_Hat Wearing_Lipstick Wearing_Necklace Wearing_Necktie'.split(),
    'na': ['Mouth_Slightly_Open'],
}
ATTR = ('5_o_Clock_Shadow Arched_Eyebrows Attractive Bags_Under_Eyes Bald Bangs Big_Lips Big_Nose Black_Hair Blond_Hair Blurry Brown_Hair '
        'Bushy_Eyebrows Chubby Double_Chin Eyeglasses Goatee Gray_Hair Heavy_Makeup High_Cheekbones Male Mouth_Slightly_Open Mustache '
        'Narrow_Eyes No_Beard Oval_Face Pale_Skin Pointy_Nose Receding_Hairline Rosy_Cheeks Sideburns Smiling Straight_Hair Wavy_Hair '
        'Wearing_Earrings Wearing_Hat Wearing_Lipstick Wearing_Necklace Wearing_Necktie Young').split()
ATTR_VERB = {a: v for v, names in _VERBS.items() for a in names}
assert len(ATTR) == 40 and set(ATTR) == set(ATTR_VERB)


def _plain(attr):
    return attr.replace('No_', '').replace('Wearing_', '').replace('_', ' ').lower()


NAME = [_plain(a) for a in ATTR]
NAME[0] = "5 o'clock shadow"  # pcfg.py:27
GET_NAME = dict(zip(ATTR, NAME))
NEGATE_IDX = [i for i, a in enumerate(ATTR) if a.startswith('No_')]
GENDER_IDX = ATTR.index('Male')
HAIR_COLOURS = ['Black_Hair', 'Blond_Hair', 'Brown_Hair', 'Gray_Hair']


def generate_phrase(male=(True, 0.5), attr=('is', 'male'), rng=None):
    """One sentence about one attribute group.  male = (is_male, probability of the pronoun form); attr = (verb, text)."""
    rng = rng or _random
    is_male, p_pronoun = male
    if rng.random() > p_pronoun:
        det = rng.choice(['a', 'this'])
        if rng.random() < 0.75:
            noun = rng.choice(['male', 'man'] if is_male else ['female', 'woman'])
        else:
            noun = 'person'
        subject = f'{det} {noun}'
    else:
        subject = 'he' if is_male else 'she'
    verb, what = attr
    if verb == 'wear':
        verb_text = rng.choice(['wears', 'is wearing'])
    elif verb in ('is', 'has'):
        verb_text = verb
    else:
        raise ValueError(f'no sentence form for verb {verb!r}')
    sentence = f'{subject} {verb_text} {what}'
    return sentence[0].upper() + sentence[1:] + '.'


def _take_group(names, rng, p2=0.9, p3=0.85):
    """Pops one to three names off the front of `names` and joins them (pcfg.py:104-117)."""
    group = [names.pop(0)]
    if names and rng.random() < p2:
        group.append(names.pop(0))
    if names and rng.random() < p3:
        group.append(names.pop(0))
    if len(group) == 1:
        return group[0]
    return ', '.join(group[:-1]) + ' and ' + group[-1]


def generate(pred, n=10, rng=None):
    """pred: 40 booleans (label-file order) -> n descriptions, each one sentence per attribute group.  The grouping is drawn
    once and shared by the n descriptions, as in pcfg.py:80-102; 'Male' only selects the pronoun."""
    rng = rng or _random
    pred = [bool(p) for p in pred]
    if len(pred) != 40:
        raise ValueError(f'expected 40 attribute flags, got {len(pred)}')
    for i in NEGATE_IDX:  # a 'No_Beard' flag is stored negated: 1 = has a beard
        pred[i] = not pred[i]
    present = [a for a, p in zip(ATTR, pred) if p]
    rng.shuffle(present)
    pools = {v: [GET_NAME[a] for a in present if ATTR_VERB[a] == v and a != 'Male'] for v in ('wear', 'has', 'is')}
    groups = []
    while any(pools.values()):
        verbs = [v for v in ('wear', 'has', 'is')]
        weights = [len(pools[v]) for v in verbs]
        verb = rng.choices(verbs, weights=weights)[0]  # a verb with probability proportional to what is left of it
        groups.append((verb, _take_group(pools[verb], rng)))
    out = []
    for _ in range(n):
        parts = [generate_phrase((pred[GENDER_IDX], 0.5 if gi == 0 else 0.85), g, rng) for gi, g in enumerate(groups)]
        out.append(' '.join(parts))
    return out


def mutual_exclusive(pred, subset, rng=None):
    """At most one attribute of `subset` stays set (pcfg.py:184-195)."""
    rng = rng or _random
    idx = [ATTR.index(a) for a in subset]
    if sum(bool(pred[i]) for i in idx) > 1:
        keep = rng.randrange(len(idx))
        for j, i in enumerate(idx):
            pred[i] = j == keep
    return pred


def generate_random_sentences(n_attr=8, n_sent=16, rng=None):
    """Random attribute sets -> one description each (pcfg.py:198-214: the prompts of the attribute-conditioned demos)."""
    rng = rng or _random
    muted = ('Attractive', 'Brown_Hair', 'Mouth_Slightly_Open', 'Blurry', 'Smiling')
    out = []
    for _ in range(n_sent):
        pred = [rng.random() < n_attr / 40 for _ in range(40)]
        pred = mutual_exclusive(pred, HAIR_COLOURS, rng)
        pred[GENDER_IDX] = rng.random() < 0.5
        for a in muted:
            pred[ATTR.index(a)] = False
        out += generate(pred, 1, rng)
    return out
