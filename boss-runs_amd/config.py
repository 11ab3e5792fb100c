"""Configuration objects with the reference's field names (boss/config.py:24-69), restricted to
what the decision-update path reads.  Plain dataclasses: no TOML/CLI layer (out of scope)."""
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class GeneralConfig:
    name: str = 'boss'
    ref: Optional[str] = None
    mmi: Optional[str] = None
    toml_readfish: Optional[str] = None
    wait: int = 60
    barcodes: Optional[List[str]] = None


@dataclass
class OptionalConfig:
    reject_refs: Optional[str] = None
    ploidy: int = 1
    bucket_threshold: int = 5


@dataclass
class LiveConfig:
    """boss/config.py:33-37 (read by Boss.launch_live_components / _init_live: out of scope here,
    kept so that reference-shaped callers find the section)."""
    device: Optional[str] = None
    host: str = 'localhost'
    port: int = 9502
    data_wait: int = 100


@dataclass
class SimulationConfig:
    """boss/config.py:54-62."""
    fq: Optional[str] = None
    batchsize: int = 4000
    maxb: int = 400
    binit: int = 5
    dumptime: int = 200000000
    paf_full: Optional[str] = None
    paf_trunc: Optional[str] = None
    accept_unmapped: bool = False


@dataclass
class GpuConfig:
    """Additions of this build (no reference counterpart)."""
    device: int = 0
    track_entropy: bool = True
    # "npz": masks/boss.npz as the reference writes it; "bits": masks/boss.bits (masks.py, 8x
    # smaller, mapped by the consumer); "both"
    mask_format: str = "npz"
    # The device picks the threshold bin from EXACT sums, the reference from 12-chunk float sums (sequences.py:609-636); both
    # agree unless the argmax of cs_u / cs_t sits within rounding of a tie.  Below this relative margin between the best and
    # the second-best ratio (bossx_update_result.argmax_margin) the update re-derives the threshold on the host in the
    # reference's own summation order (runs.reference_order_threshold) and re-forms the masks with it.  0 disables.
    tie_margin: float = 1e-9


@dataclass
class BossConfig:
    general: GeneralConfig = field(default_factory=GeneralConfig)
    live: LiveConfig = field(default_factory=LiveConfig)
    optional: OptionalConfig = field(default_factory=OptionalConfig)
    simulation: SimulationConfig = field(default_factory=SimulationConfig)
    gpu: GpuConfig = field(default_factory=GpuConfig)
