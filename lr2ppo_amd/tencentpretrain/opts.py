"""Command-line flag groups with the names and defaults of the reference's tencentpretrain/opts.py
(finetune_opts :129-151 and the groups it pulls in, tokenizer_opts :176-197, adv_opts :221-233), so the
LR2PPO launchers' argument lists (ppo.sh:42-63) parse unchanged.  Table-driven; flags that only concern
model families outside the hot path (audio front-end, deepspeed, apex fp16) are accepted and ignored.
"""

_EMB = ["word", "pos", "seg", "sinusoidalpos", "patch", "speech", "word_patch", "dual"]
_ENC = ["transformer", "rnn", "lstm", "gru", "birnn", "bilstm", "bigru", "gatedcnn", "dual"]
_TOK = ["bert", "bpe", "char", "space", "xlmroberta", "image", "text_image", "virtual"]
_LVL = ["ERROR", "INFO", "DEBUG", "NOTSET"]
_FLAG = dict(action="store_true")

_PATHS = [
    ("--pretrained_model_path", dict(default=None, type=str)),
    ("--output_model_path", dict(default="models/finetuned_model.bin", type=str)),
    ("--train_path", dict(type=str, required=False)),
    ("--dev_path", dict(type=str, required=False)),
    ("--test_path", dict(default=None, type=str)),
    ("--config_path", dict(default="models/bert/base_config.json", type=str)),
]
_MODEL = [
    ("--embedding", dict(choices=_EMB, default="word", nargs="+")),
    ("--tgt_embedding", dict(choices=_EMB, default="word", nargs="+")),
    ("--max_seq_length", dict(type=int, default=512)),
    ("--relative_position_embedding", _FLAG), ("--share_embedding", _FLAG), ("--remove_embedding_layernorm", _FLAG),
    ("--factorized_embedding_parameterization", _FLAG),
    ("--encoder", dict(choices=_ENC, default="transformer")),
    ("--decoder", dict(choices=[None, "transformer"], default=None)),
    ("--mask", dict(choices=["fully_visible", "causal", "causal_with_prefix"], default="fully_visible")),
    ("--layernorm_positioning", dict(choices=["pre", "post"], default="post")),
    ("--feed_forward", dict(choices=["dense", "gated"], default="dense")),
    ("--relative_attention_buckets_num", dict(type=int, default=32)),
    ("--remove_attention_scale", _FLAG), ("--remove_transformer_bias", _FLAG),
    ("--layernorm", dict(choices=["normal", "t5"], default="normal")),
    ("--bidirectional", _FLAG), ("--parameter_sharing", _FLAG), ("--has_residual_attention", _FLAG),
    ("--has_lmtarget_bias", _FLAG),
    ("--target", dict(choices=["sp", "lm", "mlm", "bilm", "cls", "clr"], default="mlm", nargs="+")),
    ("--tie_weights", _FLAG),
    ("--pooling", dict(choices=["mean", "max", "first", "last"], default="first")),
]
_VISION = [
    ("--image_height", dict(type=int, default=256)), ("--image_width", dict(type=int, default=256)),
    ("--patch_size", dict(type=int, default=16)), ("--channels_num", dict(type=int, default=3)),
    ("--image_preprocess", dict(type=str, default=["crop", "normalize"], nargs="+")),
]
_AUDIO = [
    ("--sampling_rate", dict(type=int, default=16000)),
    ("--audio_preprocess", dict(type=str, default=["normalize_means", "normalize_vars", "ceptral_normalize"], nargs="+")),
    ("--max_audio_frames", dict(type=int, default=6000)), ("--conv_layers_num", dict(type=int, default=2)),
    ("--audio_feature_size", dict(type=int, default=80)), ("--conv_channels", dict(type=int, default=1024)),
    ("--conv_kernel_sizes", dict(type=int, default=[5, 5], nargs="+")),
]
_OPTIM = [
    ("--learning_rate", dict(type=float, default=2e-5)), ("--warmup", dict(type=float, default=0.1)),
    ("--decay", dict(type=float, default=0.5)), ("--fp16", _FLAG),
    ("--fp16_opt_level", dict(choices=["O0", "O1", "O2", "O3"], default="O1")),
    ("--optimizer", dict(choices=["adamw", "adafactor"], default="adamw")),
    ("--scheduler", dict(choices=["linear", "cosine", "cosine_with_restarts", "polynomial", "constant",
                                  "constant_with_warmup", "inverse_sqrt", "tri_stage"], default="linear")),
]
_TRAIN = [
    ("--batch_size", dict(type=int, default=32)), ("--seq_length", dict(type=int, default=128)),
    ("--max_imgs", dict(type=int, default=32)), ("--visual_feat_dim", dict(type=int, default=-1)),
    ("--dropout", dict(type=float, default=0.1)), ("--epochs_num", dict(type=int, default=3)),
    ("--report_steps", dict(type=int, default=100)), ("--seed", dict(type=int, default=7)),
]
_LOG = [
    ("--log_path", dict(type=str, default=None)), ("--log_level", dict(choices=_LVL, default="INFO")),
    ("--log_file_level", dict(choices=_LVL, default="INFO")),
]
_TOKENIZER = [
    ("--tokenizer", dict(choices=_TOK, default="bert")), ("--vocab_path", dict(default=None, type=str)),
    ("--merges_path", dict(default=None, type=str)), ("--spm_model_path", dict(default=None, type=str)),
    ("--do_lower_case", dict(choices=["true", "false"], default="true")),
    ("--vqgan_model_path", dict(default=None, type=str)), ("--vqgan_config_path", dict(default=None, type=str)),
]
_ADV = [
    ("--use_adv", _FLAG), ("--adv_type", dict(choices=["fgm", "pgd"], default="fgm")),
    ("--fgm_epsilon", dict(type=float, default=1e-6)), ("--pgd_k", dict(type=int, default=3)),
    ("--pgd_epsilon", dict(type=float, default=1.0)), ("--pgd_alpha", dict(type=float, default=0.3)),
]


def _add(parser, table):
    seen = {a for act in parser._actions for a in act.option_strings}
    for flag, kw in table:
        if flag not in seen:
            parser.add_argument(flag, **kw)


def model_opts(parser):
    _add(parser, _MODEL + _VISION + _AUDIO)


def vision_opts(parser):
    _add(parser, _VISION)


def log_opts(parser):
    _add(parser, _LOG)


def optimization_opts(parser):
    _add(parser, _OPTIM)


def training_opts(parser):
    _add(parser, _TRAIN + _LOG)


def finetune_opts(parser):
    _add(parser, _PATHS + _MODEL + _VISION + _AUDIO + _OPTIM + _TRAIN + _LOG)


def tokenizer_opts(parser):
    _add(parser, _TOKENIZER)


def adv_opts(parser):
    _add(parser, _ADV)
