"""``kvarq.log`` -- the logger the engine reports through (csrc/workhorse.c:1605-1609 resolves ``kvarq.log.lo.log``)."""
from kvarq_amd.log import lo      # noqa: F401
