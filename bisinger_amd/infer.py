"""Inference harness — the reference's entry points for this path
(inference/m4singer/bisinger/a-lang-esm-style-ori-shift.py:152-635, base_svs_infer.py:18-357), kept by name:

    BaseSVSInfer / DiffSingerE2EInfer(hparams, device=None)
        build_model, build_vocoder, preprocess_input, input_to_batch, forward_model, run_vocoder,
        postprocess_output, infer_once, example_run, infer_from_json
    + forward_batch(items): true batched generation (the reference harness is B = 1)

In scope: everything from the phoneme/note item dict to the waveform.  Out of scope: the lyric front-end
(pinyin / CMU G2P, spaCy; hard-coded lexicon paths a-*.py:165-172) — ``input_type='word'`` raises; use
``input_type='phoneme'`` (ph_seq / note_seq / note_dur_seq / is_slur_seq / lang_seq).
"""
import json
import os
import re

import numpy as np
import torch

from .ckpt import latest_ckpt, load_ckpt
from .diffnet import DIFF_DECODERS
from .diffusion import GaussianDiffusion
from .hparams import hparams, set_hparams
from .text_encoder import TokenTextEncoder

_NOTE_BASE = {'C': 0, 'D': 2, 'E': 4, 'F': 5, 'G': 7, 'A': 9, 'B': 11}


def note_to_midi(note):
    """'C4' -> 60, 'F#3' / 'Gb3' -> 54 (the subset of librosa.note_to_midi the harness needs, a-*.py:468-471)."""
    m = re.fullmatch(r'([A-Ga-g])([#b♯♭!]*)(-?\d+)', note.strip())
    if not m:
        raise ValueError(f'bad note name {note!r}')
    acc = sum(1 if c in '#♯' else -1 for c in m.group(2))
    return 12 * (int(m.group(3)) + 1) + _NOTE_BASE[m.group(1).upper()] + acc


def save_wav(wav, path, sr, norm=False):
    """utils/audio.py:13 — 16-bit PCM."""
    from scipy.io import wavfile
    wav = np.asarray(wav, dtype=np.float32)
    if norm:
        wav = wav / max(np.abs(wav).max(), 1e-8)
    wavfile.write(path, sr, (np.clip(wav, -1, 1) * 32767).astype(np.int16))


class BaseSVSInfer:
    def __init__(self, hparams, device=None):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError('bisinger_amd runs on an MI355X; there is no CPU path')
            device = 'cuda'
        self.hparams = hparams
        self.device = device
        with open(os.path.join(hparams['binary_data_dir'], 'phone_set.json')) as f:
            self.phone_list = json.load(f)
        self.ph_encoder = TokenTextEncoder(None, vocab_list=self.phone_list, replace_oov=',')
        with open(os.path.join(hparams['binary_data_dir'], 'spk_map.json')) as f:
            self.spk_map = json.load(f)
        self.model = self.build_model()
        self.model.eval()
        self.model.to(self.device)
        self.vocoder = self.build_vocoder()
        self.vocoder.eval()
        self.vocoder.to(self.device)

    def build_model(self):
        raise NotImplementedError

    def forward_model(self, inp):
        raise NotImplementedError

    def build_vocoder(self):
        from .hifigan import HifiGanGenerator
        base_dir = hparams['vocoder_ckpt']
        ckpt = latest_ckpt(base_dir)
        assert ckpt is not None, f'no HiFi-GAN checkpoint in {base_dir}'
        print('| load HifiGAN: ', ckpt)
        config = set_hparams(f'{base_dir}/config.yaml', global_hparams=False, print_hparams=False)
        config.setdefault('use_pitch_embed', False)
        vocoder = HifiGanGenerator(config)
        vocoder.load_state_dict(torch.load(ckpt, map_location='cpu')['state_dict']['model_gen'], strict=True)
        vocoder = vocoder.eval().to(self.device)
        vocoder.remove_weight_norm()
        return vocoder

    def run_vocoder(self, c, **kwargs):
        """c [B,T,80] -> [1, B*T*hop] (a-*.py:209-218; the reference flattens, B is 1 there)."""
        c = c.transpose(2, 1)
        f0 = kwargs.get('f0')
        if f0 is not None and hparams.get('use_nsf'):
            return self.vocoder(c, f0, seed=int(kwargs.get('seed', hparams.get('seed', 1234)))).view(-1)[None]
        return self.vocoder(c).view(-1)[None]

    # ------------------------------------------------------------------ front end (tensor level)
    def preprocess_word_level_input(self, inp):
        raise NotImplementedError('lyric G2P front-end is out of scope (needs the reference authors\' lexicons); '
                                  "pass input_type='phoneme'")

    def preprocess_phoneme_level_input(self, inp):
        ph_seq = inp['ph_seq']
        note_lst = inp['note_seq'].split()
        midi_dur_lst = inp['note_dur_seq'].split()
        is_slur = [int(float(x)) for x in inp['is_slur_seq'].split()]
        lang = [int(float(x)) for x in inp['lang_seq'].split()]
        n = len(ph_seq.split())
        if not (len(note_lst) == n == len(midi_dur_lst) == len(is_slur) == len(lang)):
            print("The number of phonemes doesn't match the number of notes.")
            return None
        return ph_seq, note_lst, midi_dur_lst, is_slur, lang, int(inp.get('speechsing', 1))

    def preprocess_input(self, inp, input_type='word'):
        item_name = inp.get('item_name', '<ITEM_NAME>')
        spk_id = self.spk_map[inp.get('spk_name', 'Tenor-1')]
        if input_type == 'word':
            ret = self.preprocess_word_level_input(inp)
        elif input_type == 'phoneme':
            ret = self.preprocess_phoneme_level_input(inp)
        else:
            print('Invalid input type.')
            return None
        if not ret:
            return None
        ph_seq, note_lst, midi_dur_lst, is_slur, lang, speechsing = ret
        try:
            midis = [note_to_midi(x.split('/')[0]) if x != 'rest' else 0 for x in note_lst]
            midi_dur_lst = [float(x) for x in midi_dur_lst]
        except Exception as e:
            print(e)
            print('Invalid Input Type.')
            return None
        item = {'item_name': item_name, 'text': inp.get('text', ''), 'ph': ph_seq, 'spk_id': spk_id,
                'ph_token': self.ph_encoder.encode(ph_seq), 'pitch_midi': np.asarray(midis),
                'midi_dur': np.asarray(midi_dur_lst), 'is_slur': np.asarray(is_slur), 'lang': np.asarray(lang),
                'speechsing': speechsing}
        item['ph_len'] = len(item['ph_token'])
        return item

    def input_to_batch(self, item):
        return self.collate([item])

    def collate(self, items):
        """Batch of items, padded with id 0 (utils/__init__.py:45-60 collate_1d semantics)."""
        mf = hparams['max_frames']
        n = max(len(it['ph_token']) for it in items)

        def pad(key, dtype):
            out = torch.zeros(len(items), n, dtype=dtype)
            for i, it in enumerate(items):
                v = torch.as_tensor(np.asarray(it[key]))[:mf]
                out[i, :len(v)] = v.to(dtype)
            return out.to(self.device)
        return {
            'item_name': [it['item_name'] for it in items], 'text': [it['text'] for it in items], 'ph': [it['ph'] for it in items],
            'txt_tokens': pad('ph_token', torch.long),
            'txt_lengths': torch.LongTensor([len(it['ph_token']) for it in items]).to(self.device),
            'spk_ids': torch.LongTensor([it['spk_id'] for it in items]).to(self.device),
            'pitch_midi': pad('pitch_midi', torch.long), 'midi_dur': pad('midi_dur', torch.float32),
            'is_slur': pad('is_slur', torch.long), 'lang': pad('lang', torch.long),
            'speechsing': torch.LongTensor([it['speechsing'] for it in items]).to(self.device),
        }

    def postprocess_output(self, output):
        return output

    def infer_once(self, inp):
        inp = self.preprocess_input(inp, input_type=inp['input_type'] if inp.get('input_type') else 'word')
        output = self.forward_model(inp)
        return self.postprocess_output(output)

    @classmethod
    def example_run(cls, inp):
        set_hparams(print_hparams=False)
        infer_ins = cls(hparams)
        out = infer_ins.infer_once(inp)
        os.makedirs('infer_out', exist_ok=True)
        f_name = inp['spk_name'] + ' | ' + inp.get('text', inp.get('item_name', 'out'))
        save_wav(out, f'infer_out/{f_name}.wav', hparams['audio_sample_rate'])

    @classmethod
    def infer_from_json(cls, inp_fn, save_path, bpm=None):
        set_hparams(print_hparams=True)
        infer_ins = cls(hparams)
        inps = json.load(open(inp_fn))
        os.makedirs(save_path, exist_ok=True)
        for inp in inps:
            inp.setdefault('spk_name', f'{inp_fn.split("/")[-2]}-1')
            out = infer_ins.infer_once(inp)
            f_name = str(inp.get('id', inp.get('item_name', 'item'))) + '|' + inp['spk_name'] + '|' + inp.get('text', '')
            save_wav(out, f'{save_path}/{f_name}.wav', hparams['audio_sample_rate'])


class DiffSingerE2EInfer(BaseSVSInfer):
    def build_model(self):
        model = GaussianDiffusion(phone_encoder=self.ph_encoder, out_dims=hparams['audio_num_mel_bins'],
                                  denoise_fn=DIFF_DECODERS[hparams['diff_decoder_type']](hparams),
                                  timesteps=hparams['timesteps'], K_step=hparams['K_step'],
                                  loss_type=hparams['diff_loss_type'], spec_min=hparams['spec_min'],
                                  spec_max=hparams['spec_max'])
        model.eval()
        load_ckpt(model, hparams['work_dir'], 'model')
        if hparams.get('pe_enable'):
            from .pe import PitchExtractor
            self.pe = PitchExtractor().to(self.device)
            load_ckpt(self.pe, hparams['pe_ckpt'], 'model', strict=True)
            self.pe.eval()
        return model

    def _generate(self, sample, seed=None):
        with torch.no_grad():
            output = self.model(sample['txt_tokens'], spk_embed=sample.get('spk_ids'), ref_mels=None, infer=True,
                                pitch_midi=sample['pitch_midi'], midi_dur=sample['midi_dur'], is_slur=sample['is_slur'],
                                lang=sample['lang'], speechsing=sample['speechsing'], seed=seed)
        return output

    def forward_model(self, inp, seed=None):
        sample = self.input_to_batch(inp)
        output = self._generate(sample, seed)
        mel_out = output['mel_out']
        f0_pred = self.pe(mel_out)['f0_denorm_pred'] if hparams.get('pe_enable') else None      # a-*.py:629-632
        wav_out = self.run_vocoder(mel_out, f0=f0_pred)
        return wav_out.cpu().numpy()[0]

    def estimate_frames(self, item):
        """Frames an item will take, before any model runs (the true count is the duration predictor's): consecutive phonemes
        that share a note (same pitch and duration entry) count once — a-*.py:468-483 repeats a note's duration for each of
        its phonemes.  Only used to order and budget buckets; never for shapes."""
        d = np.asarray(item['midi_dur'], dtype=np.float64)
        p = np.asarray(item['pitch_midi'])
        if len(d) == 0:
            return 0
        new = np.ones(len(d), dtype=bool)
        new[1:] = (d[1:] != d[:-1]) | (p[1:] != p[:-1])
        return int(np.ceil(d[new].sum() * hparams['audio_sample_rate'] / hparams['hop_size']))

    def _generate_wavs(self, items, seed):
        """One padded batch -> (list of trimmed waveforms, frames per item, padded frame count T)."""
        sample = self.collate(items)
        output = self._generate(sample, seed)
        mel, mel2ph = output['mel_out'], output['mel2ph']
        hop = int(np.prod(self.vocoder.h['upsample_rates']))
        if hparams.get('use_nsf'):                                   # the shipped M4Singer set-up: PitchExtractor f0 -> NSF source
            f0 = self.pe(mel)['f0_denorm_pred'] if hparams.get('pe_enable') else output.get('f0_denorm')
            assert f0 is not None, 'use_nsf needs an f0: enable pe_enable (a-*.py:629-632)'
            wav = self.vocoder(mel.transpose(2, 1), f0, seed=int(hparams.get('seed', 1234) if seed is None else seed))[:, 0]
        else:
            wav = self.vocoder(mel.transpose(2, 1))[:, 0]
        n_frames = (mel2ph > 0).sum(-1).tolist()
        return [wav[i, :n * hop].cpu().numpy() for i, n in enumerate(n_frames)], n_frames, int(mel.shape[1])

    def forward_batch(self, items, seed=None, max_frames=None, max_sentences=None):
        """Batched generation: list of items -> list of 1-D waveforms (in the order given), each trimmed to its utterance.

        With ``max_frames`` (budget on padded frames per batch: rows x longest row) and/or ``max_sentences`` the items are
        length-bucketed the way the reference batches its datasets (utils/__init__.py:90-143 batch_by_size over
        size-ordered indices): sorted by estimated length, packed greedily while (n + 1) x longest <= max_frames and
        n < max_sentences; each bucket is one padded batch.  Without either, all items form one batch.

        NB (reference quirk kept): ESM attends over the batch axis (common_layers.py:853), so a row's result depends on
        the rows it is batched with — the unit of reproducibility (and of the oracle in the tests) is the BUCKET.
        ``self.last_batch_stats`` reports the buckets and the padded-frame waste with and without bucketing."""
        est = [self.estimate_frames(it) for it in items]
        if max_frames is None and max_sentences is None:
            buckets = [list(range(len(items)))]
        else:
            buckets = bucket_by_size(est, max_frames, max_sentences)
        wavs, frames = [None] * len(items), [0] * len(items)
        padded = 0
        for b in buckets:
            w, nf, T = self._generate_wavs([items[i] for i in b], seed)
            padded += T * len(b)
            for i, wi, ni in zip(b, w, nf):
                wavs[i], frames[i] = wi, ni
        real = int(sum(frames))
        one = max(frames) * len(items) if items else 0
        self.last_batch_stats = {
            'buckets': buckets, 'estimated_frames': est, 'frames': frames, 'real_frames': real, 'padded_frames': padded,
            'waste': 1.0 - real / max(padded, 1), 'padded_frames_single_batch': one, 'waste_single_batch': 1.0 - real / max(one, 1)}
        return wavs


def bucket_by_size(lengths, max_frames=None, max_sentences=None):
    """Length bucketing with the semantics of the reference's batch_by_size (utils/__init__.py:90-143) applied to
    size-sorted indices: walk the items from longest to shortest and close the current bucket when one more row would make
    rows x longest exceed ``max_frames`` or the row count exceed ``max_sentences``.  Returns lists of indices into
    ``lengths``; every index appears exactly once.  An item longer than ``max_frames`` raises, as the reference asserts."""
    max_frames = float('inf') if max_frames is None else max_frames
    max_sentences = float('inf') if max_sentences is None else max_sentences
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    buckets, cur, longest = [], [], 0
    for i in order:
        if lengths[i] > max_frames:
            raise ValueError(f'item {i} of {lengths[i]} frames exceeds max_frames={max_frames}')
        new_longest = max(longest, lengths[i])
        if cur and ((len(cur) + 1) * new_longest > max_frames or len(cur) + 1 > max_sentences):
            buckets.append(cur)
            cur, new_longest = [], lengths[i]
        cur.append(i)
        longest = new_longest
    if cur:
        buckets.append(cur)
    return buckets
