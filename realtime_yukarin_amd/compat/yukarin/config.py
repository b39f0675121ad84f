"""`yukarin.config.create_from_json` ([MEM] schema; called at
/root/reference/check.py:31 and realtime_voice_conversion/converter/yukarin_converter.py:39).
Parses tolerantly: unknown keys are ignored, the attribute paths the reference reads are required."""
import json
import os
from pathlib import Path
from typing import List, NamedTuple, Optional, Union

from .param import AcousticParam


class DatasetConfig(NamedTuple):
    acoustic_param: AcousticParam
    in_features: List[str]
    out_features: List[str]
    extra: dict


class ModelConfig(NamedTuple):
    in_channels: int
    out_channels: int
    generator_base_channels: int
    generator_extensive_layers: int
    glu_generator: bool = False
    extra: Optional[dict] = None


class Config(NamedTuple):
    dataset: DatasetConfig
    model: ModelConfig
    raw: dict


def _acoustic_param(d: dict) -> AcousticParam:
    known = {k: v for k, v in d.items() if k in AcousticParam._fields}
    return AcousticParam(**known)


def create_from_dict(d: dict) -> Config:
    ds, md = d['dataset'], d['model']
    feats_default = ['mc']
    dataset = DatasetConfig(
        acoustic_param=_acoustic_param(ds.get('acoustic_param', ds.get('param', {}))),
        in_features=list(ds.get('in_features', ds.get('features', feats_default))),
        out_features=list(ds.get('out_features', ds.get('features', feats_default))),
        extra={k: v for k, v in ds.items() if k not in ('acoustic_param', 'in_features', 'out_features')},
    )
    model = ModelConfig(
        in_channels=int(md['in_channels']),
        out_channels=int(md['out_channels']),
        generator_base_channels=int(md.get('generator_base_channels', 64)),
        generator_extensive_layers=int(md.get('generator_extensive_layers', 8)),
        glu_generator=bool(md.get('glu_generator', False)),
        extra={k: v for k, v in md.items()},
    )
    if model.glu_generator and os.environ.get('RY_ALLOW_UNVERIFIED_GLU', '0') != '1':
        # The gated predictor class lives in the un-vendored `yukarin` package; what is built here is this repository's READING of it
        # ([MEM], realtime_yukarin_amd/netspec.py: every conv + BN block computes twice the channels and is gated, a * sigmoid(b)).
        # The strict K-list / shape validation of the weight loader still decides whether a real model file fits that reading.
        raise NotImplementedError('glu_generator: the gated stage-1 predictor is built from an UNVERIFIED reading of the upstream architecture; '
                                  'set RY_ALLOW_UNVERIFIED_GLU=1 to load it (the weight file must match its K-list exactly)')
    return Config(dataset=dataset, model=model, raw=d)


def create_from_json(s: Union[str, Path]) -> Config:
    with open(str(s)) as f:
        return create_from_dict(json.load(f))
