"""Import alias: the package directory is named `zk-nullifier-sig_amd/` (not a valid Python identifier), so this
shim package extends its search path to that directory and re-exports it.  `import zk_nullifier_sig_amd as plume`."""
from pathlib import Path as _Path

__path__.append(str(_Path(__file__).resolve().parent.parent / "zk-nullifier-sig_amd"))

from .capi import Engine, PlumeHipError, default_engine, library_path  # noqa: E402,F401
from .plume import (  # noqa: E402,F401
    DST,
    AffinePoint,
    NonZeroScalar,
    PlumePanic,
    PlumeSignature,
    PlumeSignaturePrivate,
    PlumeSignaturePublic,
    PlumeSignatureV1Fields,
    PlumeSigner,
    PlumeVersion,
    SecretKey,
    SignatureError,
    circuit_inputs,
    sign,
    sign_with_r,
    verify_non_zk,
)
from . import nullifier_set  # noqa: E402,F401
