"""Framework-free restatement of the hot-path HTTP handlers.

``compute_bytes`` is the body of the reference's ``views.compute`` (main/views.py:30-42) with the
Django request/response objects peeled off: request bytes in, ``(status, bytes)`` out.  Any
exception becomes status 400 with ``str(e)`` as the body - the reference's error convention.
``description`` / ``contents`` restate views.py:17-28 the same way.  A Django (or ASGI) view is a
three-line wrapper around these; see INTEGRATION.md.
"""
from __future__ import annotations

import json
import logging
from typing import Mapping, Optional, Tuple

from .context import Context, context
from .message import Request, Response

logger = logging.getLogger(__name__)


def compute_bytes(body: bytes, ctx: Optional[Context] = None) -> Tuple[int, bytes]:
    try:
        req = Request()
        req.decode(body)
        if logger.isEnabledFor(logging.DEBUG):
            logger.debug("%s", req.graph)
        (ctx or context()).compute(req.graph)
        if logger.isEnabledFor(logging.DEBUG):
            logger.debug("%s", req.graph)
        return 200, Response(req.graph).encode()
    except Exception as e:
        logger.error(e)
        return 400, str(e).encode()


def description(name: str, params: Mapping[str, str], ctx: Optional[Context] = None) -> Tuple[int, bytes]:
    try:
        return 200, json.dumps((ctx or context()).get_node(name).io(params)).encode()
    except Exception as e:
        return 400, str(e).encode()


def contents(name: str, params: Mapping[str, str], ctx: Optional[Context] = None) -> Tuple[int, bytes]:
    try:
        return 200, (ctx or context()).get_node(name).contents(params).encode()
    except Exception as e:
        return 400, str(e).encode()
