"""`become_yukarin.config.sr_config.create_from_json` ([MEM] schema; called at
/root/reference/check.py:32 and realtime_voice_conversion/converter/yukarin_converter.py:49)."""
import json
from pathlib import Path
from typing import NamedTuple, Optional, Union

from ..param import AcousticFeatureParam, Param, VoiceParam


class SRDatasetConfig(NamedTuple):
    param: Param
    extra: dict


class SRModelConfig(NamedTuple):
    generator_base_channels: int
    generator_extensive_layers: int
    extra: Optional[dict] = None


class SRConfig(NamedTuple):
    dataset: SRDatasetConfig
    model: SRModelConfig
    raw: dict


def _param(d) -> Param:
    if not isinstance(d, dict):
        return Param()
    vp = {k: v for k, v in d.get('voice_param', {}).items() if k in VoiceParam._fields}
    ap = {k: v for k, v in d.get('acoustic_feature_param', {}).items() if k in AcousticFeatureParam._fields}
    return Param(voice_param=VoiceParam(**vp), acoustic_feature_param=AcousticFeatureParam(**ap))


def create_from_dict(d: dict) -> SRConfig:
    ds, md = d.get('dataset', {}), d.get('model', {})
    return SRConfig(
        dataset=SRDatasetConfig(param=_param(ds.get('param')), extra={k: v for k, v in ds.items() if k != 'param'}),
        model=SRModelConfig(generator_base_channels=int(md.get('generator_base_channels', 64)),
                            generator_extensive_layers=int(md.get('generator_extensive_layers', 8)), extra=dict(md)),
        raw=d,
    )


def create_from_json(s: Union[str, Path]) -> SRConfig:
    with open(str(s)) as f:
        return create_from_dict(json.load(f))
