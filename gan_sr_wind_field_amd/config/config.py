"""INI configuration, drop-in for the reference's ``config/config.py``.

Same public names (``Config`` and the per-section ``*Config`` attribute bags),
same attribute names and ``None``-for-bare-key semantics, same ``str(cfg)`` /
``asINI()`` text (reference config/config.py:18-396), so every shipped
``config/*.ini`` and ``pretrained_models/*/config.ini`` loads unchanged.
Implemented table-driven: each section is a list of ``(key, kind)`` pairs.

Extension (optional keys, defaults keep reference behaviour):
  [DEFAULT] compute_dtype = fp32 | bf16     arithmetic type of the HIP kernels
  [DIST]    backend / bucket_mb / sync_bn   data-parallel settings (see dist.py)
"""
from __future__ import annotations

import ast
from configparser import ConfigParser
from typing import Any, List, Sequence, Tuple

_B, _I, _F, _S = "bool", "int", "float", "str"


def safe_list_from_string(text, target_type: type) -> list:
    """``ast.literal_eval`` a list literal; anything unparsable gives ``[]``
    (reference config/config.py:384-396)."""
    try:
        val = ast.literal_eval(text)
    except Exception:
        return []
    if val is None:
        return []
    return val if isinstance(val, list) else [val]


def _read(section, key: str, kind: str):
    if section.get(key) is None:
        # missing key or bare key.  (The reference crashes on a bare bool/int key, which
        # makes its own asINI() dump unloadable whenever a value was None; returning
        # None keeps every loadable file identical and makes the round trip total.)
        return [] if kind == "intlist" else None
    if kind == _B:
        return section.getboolean(key)
    if kind == _I:
        return section.getint(key)
    if kind == _F:
        return section.getfloat(key)
    if kind == "intlist":
        return safe_list_from_string(section.get(key), int)
    return section.get(key)


class IniConfig:
    """Attribute bag that prints itself back as an INI section."""

    _schema: Sequence[Tuple[str, str]] = ()

    def _load(self, section) -> None:
        for key, kind in self._schema:
            setattr(self, key, _read(section, key, kind))

    def __str__(self) -> str:
        head = "[" + type(self).__name__.upper().replace("CONFIG", "") + "]\n"
        body = "".join(f"{k}\n" if v is None else f"{k} = {v}\n" for k, v in vars(self).items())
        return head + body


class GANConfig(IniConfig):
    include_pressure: bool = True
    include_z_channel: bool = True
    include_above_ground_channel: bool = False
    number_of_z_layers: int = 10
    conv_mode: str = "3D"
    start_date = [2018, 4, 1]
    end_date = [2018, 4, 4]
    interpolate_z: bool = False
    use_D_feature_extractor_cost = False
    enable_slicing = False
    slice_size = 64
    _schema = (
        ("include_pressure", _B), ("include_z_channel", _B), ("include_above_ground_channel", _B),
        ("number_of_z_layers", _I), ("conv_mode", _S), ("start_date", "intlist"), ("end_date", "intlist"),
        ("interpolate_z", _B), ("use_D_feature_extractor_cost", _B), ("enable_slicing", _B), ("slice_size", _I),
    )

    def setGANConfig(self, section):
        self._load(section)


class EnvConfig(IniConfig):
    root_path: str = "~/GAN_SR_wind_field_"
    log_subpath: str = "/log"
    tensorboard_subpath: str = "/tensorboard_log"
    runs_subpath: str = "/runs"
    generator_load_path: str = None
    discriminator_load_path: str = None
    state_load_path: str = None
    fixed_seed: int = 2001
    this_runs_folder: str = None
    this_runs_tensorboard_folder: str = None
    _schema = (
        ("root_path", _S), ("log_subpath", _S), ("tensorboard_subpath", _S), ("runs_subpath", _S),
        ("generator_load_path", _S), ("discriminator_load_path", _S), ("state_load_path", _S), ("fixed_seed", _I),
    )

    def setEnvConfig(self, section):
        self._load(section)


class GeneratorConfig(IniConfig):
    norm_type: str = "none"
    act_type: str = "leakyrelu"
    layer_mode: str = "CNA"
    num_features: int = 64
    num_RRDB: int = 23
    num_RDB_convs: int = 5
    RDB_res_scaling: float = 0.2
    RRDB_res_scaling: float = 0.2
    in_num_ch: int = 3
    out_num_ch: int = 3
    RDB_growth_chan: int = 32
    hr_kern_size: int = 3
    weight_init_scale: float = 1.0
    lff_kern_size: int = 3
    conv_mode: str = "2D"
    use_mixed_precision: bool = True
    terrain_number_of_features: int = 16
    dropout_probability: float = 0.0
    max_norm: float = 1.0
    _schema = (
        ("norm_type", _S), ("act_type", _S), ("layer_mode", _S), ("num_features", _I), ("num_RRDB", _I),
        ("num_RDB_convs", _I), ("RDB_res_scaling", _F), ("RRDB_res_scaling", _F), ("in_num_ch", _I),
        ("out_num_ch", _I), ("RDB_growth_chan", _I), ("hr_kern_size", _I), ("weight_init_scale", _F),
        ("lff_kern_size", _I), ("conv_mode", _S), ("use_mixed_precision", _B), ("terrain_number_of_features", _I),
        ("dropout_probability", _F), ("max_norm", _F),
    )

    def setGeneratorConfig(self, section):
        self._load(section)


class DiscriminatorConfig(IniConfig):
    norm_type: str = "batch"
    act_type: str = "leakyrelu"
    layer_mode: str = "CNA"
    num_features: int = 64
    in_num_ch: int = 3
    feat_kern_size: int = 3
    weight_init_scale: float = 1.0
    conv_mode: str = "3D"
    use_mixed_precision: bool = True
    dropout_probability: float = 0.2
    _schema = (
        ("norm_type", _S), ("act_type", _S), ("layer_mode", _S), ("num_features", _I), ("in_num_ch", _I),
        ("feat_kern_size", _I), ("weight_init_scale", _F), ("conv_mode", _S), ("use_mixed_precision", _B),
        ("dropout_probability", _F),
    )

    def setDiscriminatorConfig(self, section):
        self._load(section)


class FeatureExtractorConfig(IniConfig):
    low_level_feat_layer: int = 1
    high_level_feat_layer: int = 34
    _schema = (("low_level_feat_layer", _I), ("high_level_feat_layer", _I))

    def setFeatureExtractorConfig(self, section):
        self._load(section)


class DatasetConfig(IniConfig):
    name: str = "default_dataset_name"
    mode: str = "downsampler"
    dataroot_hr: str = "default_path"
    dataroot_lr: str = "default_lr_path"
    num_workers: int = 0
    batch_size: int = 16
    data_aug_flip: bool = True
    data_aug_rot: bool = True
    _schema = (
        ("name", _S), ("mode", _S), ("dataroot_hr", _S), ("dataroot_lr", _S), ("num_workers", _I),
        ("batch_size", _I), ("data_aug_flip", _B), ("data_aug_rot", _B),
    )

    def setDatasetConfig(self, section):
        self._load(section)


class DatasetTrainConfig(DatasetConfig):
    pass


class DatasetValConfig(DatasetConfig):
    pass


class DatasetTestConfig(DatasetConfig):
    pass


class TrainingConfig(IniConfig):
    resume_training_from_save: bool = False
    learning_rate_g: float = 1e-4
    learning_rate_d: float = 1e-4
    adam_weight_decay_g: float = 0
    adam_weight_decay_d: float = 0
    adam_beta1_g: float = 0.9
    adam_beta1_d: float = 0.9
    multistep_lr: bool = True
    multistep_lr_steps: list = [50000, 100000, 200000, 300000]
    lr_gamma: float = 0.5
    train_eval_test_ratio: float = 0.8
    gan_type: str = "relativistic"
    adversarial_loss_weight: float = 5e-3
    d_g_train_ratio: int = 1
    d_g_train_period: int = 50
    pixel_criterion: str = "l1"
    pixel_loss_weight: float = 1e-1
    gradient_xy_loss_weight: float = 1e-1
    gradient_z_loss_weight: float = 1e-1
    divergence_loss_weight: float = 1e-1
    xy_divergence_loss_weight: float = 1e-1
    feature_D_loss_weight: float = 0.1
    feature_D_update_period: int = 1
    use_noisy_labels: bool = False
    use_one_sided_label_smoothing: bool = False
    flip_labels: bool = False
    use_instance_noise: bool = False
    niter: int = 25
    val_period: int = 2e3
    save_model_period: int = 2e3
    log_period: int = 1e2
    # order = the order in which the reference assigns them (it decides str(cfg))
    _schema = (
        ("resume_training_from_save", _B), ("learning_rate_g", _F), ("learning_rate_d", _F),
        ("adam_weight_decay_g", _F), ("adam_weight_decay_d", _F), ("adam_beta1_g", _F), ("adam_beta1_d", _F),
        ("multistep_lr", _B), ("multistep_lr_steps", "intlist"), ("lr_gamma", _F), ("gan_type", _S),
        ("adversarial_loss_weight", _F), ("d_g_train_ratio", _I), ("d_g_train_period", _I),
        ("pixel_criterion", _S), ("pixel_loss_weight", _F), ("gradient_xy_loss_weight", _F),
        ("gradient_z_loss_weight", _F), ("divergence_loss_weight", _F), ("xy_divergence_loss_weight", _F),
        ("feature_D_loss_weight", _F), ("use_noisy_labels", _B), ("use_one_sided_label_smoothing", _B),
        ("use_instance_noise", _B), ("flip_labels", _B), ("niter", _I), ("val_period", _I),
        ("save_model_period", _I), ("log_period", _I), ("conv_mode", _S), ("train_eval_test_ratio", _F),
        ("feature_D_update_period", _I),
    )

    def setTrainingConfig(self, section):
        self._load(section)


class DistConfig(IniConfig):
    """[DIST] (extension): data-parallel settings; absent section = defaults."""

    backend: str = "nccl"
    bucket_mb: float = 32.0
    sync_bn: bool = True
    _schema = (("backend", _S), ("bucket_mb", _F), ("sync_bn", _B))

    def setDistConfig(self, section):
        for key, kind in self._schema:
            val = _read(section, key, kind)
            if val is not None:
                setattr(self, key, val)


class Config(IniConfig):
    name: str = "default_name"
    model: str = "default_model"
    use_tensorboard_logger: bool = False
    scale: int = 4
    gpu_id: int = 0
    also_log_to_terminal: bool = True
    load_model_from_save: bool = False
    display_bar = True

    # class-level singletons, exactly like the reference (config/config.py:291-299)
    env: EnvConfig = EnvConfig()
    gan_config: GANConfig = GANConfig()
    generator: GeneratorConfig = GeneratorConfig()
    discriminator: DiscriminatorConfig = DiscriminatorConfig()
    feature_extractor: FeatureExtractorConfig = FeatureExtractorConfig()
    dataset_train: DatasetTrainConfig = DatasetTrainConfig()
    dataset_test: DatasetTestConfig = DatasetTestConfig()
    dataset_val: DatasetValConfig = DatasetValConfig()
    training: TrainingConfig = TrainingConfig()
    dist: DistConfig = DistConfig()
    compute_dtype: str = "fp32"
    is_train: bool
    is_use: bool
    is_test: bool
    is_param_search: bool
    is_download: bool
    slurm_array_id: int = 1

    def __init__(self, ini_path):
        parser = ConfigParser(allow_no_value=True)
        parser.read(ini_path)
        self.setBaseConfig(parser["DEFAULT"])
        self.gan_config.setGANConfig(parser["GAN"])
        self.env.setEnvConfig(parser["ENV"])
        self.generator.setGeneratorConfig(parser["GENERATOR"])
        self.discriminator.setDiscriminatorConfig(parser["DISCRIMINATOR"])
        self.training.setTrainingConfig(parser["TRAINING"])
        for attr, section in (("dataset_train", "DATASETTRAIN"), ("dataset_test", "DATASETTEST"),
                              ("dataset_val", "DATASETVAL")):
            if parser.has_section(section):
                getattr(self, attr).setDatasetConfig(parser[section])
            else:
                setattr(self, attr, None)
        if parser.has_section("DIST"):
            self.dist.setDistConfig(parser["DIST"])

    def setBaseConfig(self, base):
        self.name = base.get("name")
        self.model = base.get("model")
        self.use_tensorboard_logger = base.getboolean("use_tensorboard_logger")
        self.scale = base.getint("scale")
        self.also_log_to_terminal = base.getboolean("also_log_to_terminal")
        gpu = base.get("gpu_id")
        self.gpu_id = None if gpu is None or gpu.lower() == "none" else int(gpu)
        self.load_model_from_save = base.getboolean("load_model_from_save")
        self.display_bar = base.getboolean("display_bar")
        dtype = base.get("compute_dtype")
        if dtype is not None:  # optional extension key; absent -> class default, not printed
            if dtype.lower() not in ("fp32", "bf16"):
                raise ValueError(f"compute_dtype must be fp32 or bf16, not {dtype}")
            self.compute_dtype = dtype.lower()

    def asINI(self) -> str:
        return str(self)

    def __str__(self) -> str:
        out = "[DEFAULT]\n" + "".join(f"{k} = {v}\n" for k, v in vars(self).items())
        sections: List[Any] = [self.env, self.gan_config, self.generator, self.discriminator, self.training,
                               self.dataset_train, self.dataset_val, self.dataset_test]
        for sec in sections:
            if sec is not None:
                out += "\n" + str(sec)
        return out
