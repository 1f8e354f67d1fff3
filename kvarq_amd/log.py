"""``kvarq.log`` counterpart: the logger the engine reports through
(reference kvarq/log.py:31-70; the C engine resolves ``kvarq.log.lo.log`` at
import, csrc/workhorse.c:1605-1609)."""
import logging

lo = logging.getLogger('kvarq')
