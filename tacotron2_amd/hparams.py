"""TF-free hyper-parameter bag for the Tacotron 2 mel engine.

Mirrors the field set, defaults and ``k=v,k=v`` override syntax of the
reference ``create_hparams`` (reference hparams.py:5-95) without the
TensorFlow 1.x dependency (tensorflow is not installable on the MI355X image).
``n_symbols`` is the size of the reference symbol table
(reference text/symbols.py:18 -> 148 entries); the text frontend itself is
out of scope, only its cardinality sizes the embedding.
"""
import ast

N_SYMBOLS = 148  # len(text.symbols.symbols) in the reference

_DEFAULTS = dict(
    # experiment (reference hparams.py:12-22)
    epochs=500, iters_per_checkpoint=1000, seed=1234, dynamic_loss_scaling=True,
    fp16_run=False, distributed_run=False, dist_backend="nccl",
    dist_url="tcp://localhost:54321", cudnn_enabled=True, cudnn_benchmark=False,
    ignore_layers=['embedding.weight'],
    # data (reference hparams.py:27-30)
    load_mel_from_disk=False,
    training_files='filelists/ljs_audio_text_train_filelist.txt',
    validation_files='filelists/ljs_audio_text_val_filelist.txt',
    text_cleaners=['english_cleaners'],
    # audio (reference hparams.py:35-42)
    max_wav_value=32768.0, sampling_rate=22050, filter_length=1024,
    hop_length=256, win_length=1024, n_mel_channels=80, mel_fmin=0.0,
    mel_fmax=8000.0,
    # model (reference hparams.py:47-75)
    n_symbols=N_SYMBOLS, symbols_embedding_dim=512,
    encoder_kernel_size=5, encoder_n_convolutions=3, encoder_embedding_dim=512,
    n_frames_per_step=1, decoder_rnn_dim=1024, prenet_dim=256,
    max_decoder_steps=1000, gate_threshold=0.5, p_attention_dropout=0.1,
    p_decoder_dropout=0.1, attention_rnn_dim=1024, attention_dim=128,
    attention_location_n_filters=32, attention_location_kernel_size=31,
    postnet_embedding_dim=512, postnet_kernel_size=5, postnet_n_convolutions=5,
    # optimisation (reference hparams.py:80-85)
    use_saved_learning_rate=False, learning_rate=1e-3, weight_decay=1e-6,
    grad_clip_thresh=1.0, batch_size=64, mask_padding=True,
)


class HParams(object):
    """Attribute bag with ``parse("a=b,c=d")`` and ``values()`` like the
    ``tf.contrib.training.HParams`` object the reference builds."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def values(self):
        return dict(self.__dict__)

    def __contains__(self, k):
        return k in self.__dict__

    def __repr__(self):
        return "HParams(%s)" % ", ".join(
            "%s=%r" % kv for kv in sorted(self.__dict__.items()))

    @staticmethod
    def _split_top_level(s):
        out, depth, cur = [], 0, []
        for ch in s:
            if ch == '[':
                depth += 1
            elif ch == ']':
                depth -= 1
            if ch == ',' and depth == 0:
                out.append(''.join(cur))
                cur = []
            else:
                cur.append(ch)
        if cur:
            out.append(''.join(cur))
        return [x for x in (y.strip() for y in out) if x]

    @staticmethod
    def _cast(old, text):
        if isinstance(old, bool):
            low = text.strip().lower()
            if low in ('true', '1'):
                return True
            if low in ('false', '0'):
                return False
            raise ValueError("Could not parse %r as bool" % text)
        if isinstance(old, int):
            return int(text)
        if isinstance(old, float):
            return float(text)
        if isinstance(old, list):
            t = text.strip()
            if not (t.startswith('[') and t.endswith(']')):
                raise ValueError("list-valued hparam needs [a,b] syntax: %r" % text)
            items = [x.strip() for x in t[1:-1].split(',') if x.strip()]
            elem = old[0] if old else ''
            if isinstance(elem, str):
                return [x.strip('\'"') for x in items]
            return [type(elem)(ast.literal_eval(x)) for x in items]
        return text

    def parse(self, hparams_string):
        for item in self._split_top_level(hparams_string):
            if '=' not in item:
                raise ValueError("Could not parse hparam %r" % item)
            k, v = item.split('=', 1)
            k = k.strip()
            if k not in self.__dict__:
                raise ValueError("Unknown hyperparameter %r" % k)
            setattr(self, k, self._cast(getattr(self, k), v))
        return self


def create_hparams(hparams_string=None, verbose=False):
    """Create model hyperparameters. Parse nondefault from given string
    (same contract as reference hparams.py:5-95)."""
    hp = HParams(**{k: (list(v) if isinstance(v, list) else v)
                    for k, v in _DEFAULTS.items()})
    if hparams_string:
        hp.parse(hparams_string)
    if verbose:
        print('Final parsed hparams: %s' % hp.values())
    return hp
