"""Binary wire codec of ``POST /compute``: request bytes -> Graph, Graph -> response bytes.

Byte-for-byte compatible with the reference's ``main/message.py`` (Request.decode :22-73,
Response.encode :89-127, align_next :13-16); the browser side ``nodes/net_node.js:56-80,235-248``
is the normative format.  Little-endian throughout::

    u32 byte_size | u32 magic | u32 block_cnt | u32 json_size | json utf-8 | 0-pad to 4 | blocks
    block = u32 block_bytes (= 8 + 4*ndim + 4*numel) | u32 ndim | u32 dims[ndim] | f32 data[numel]

request magic 0x69babe69, json = {"nodes":[{"endpoint","params"}], "edges":[...]};
response magic 0xdeadbeef, json = [{"node": i, "channel": c}] labelling block i.

What differs from the reference is only speed: tensor payloads are mapped with
``numpy.frombuffer`` / ``torch.from_numpy`` and written through ``memoryview`` instead of walking
``array('f')`` element by element (the reference spends 3.7 s decoding one 38.5 MB block,
SURVEY 3.1).  The decoded tensors are fresh, contiguous, CPU float32 - as the reference's are.
"""
from __future__ import annotations

import ctypes
import json
import math
import sys
import logging
import struct
from typing import Dict, List, Tuple

import numpy as np
import torch

from .graph import Graph

logger = logging.getLogger(__name__)

REQUEST_MAGIC = 0x69BABE69
RESPONSE_MAGIC = 0xDEADBEEF
_HEADER = struct.Struct("<4I")
_U32x2 = struct.Struct("<2I")


def align_next(offset: int, align: int) -> int:
    """Smallest multiple of ``align`` that is >= ``offset`` (ref :13-16)."""
    return -(-offset // align) * align


def _read_block(buf: memoryview, pos: int, index: int) -> Tuple[torch.Tensor, int]:
    block_size, ndim = _U32x2.unpack_from(buf, pos)
    dims_end = pos + 8 + 4 * ndim
    dims = list(struct.unpack_from(f"<{ndim}I", buf, pos + 8))
    numel = 1
    for d in dims:
        numel *= d
    end = dims_end + 4 * numel
    if end > len(buf):
        raise AssertionError(f"tensor {index}: block runs past the end of the message")
    # ref :57 - the block's own size field must agree with ndim/dims
    assert pos + block_size == end, f"tensor {index}: block size mismatch"
    data = np.frombuffer(buf, dtype="<f4", count=numel, offset=dims_end)
    t = torch.from_numpy(data.copy()).reshape(dims)  # own the bytes; request body may be freed
    logger.info("tensor %d: size=%d, dim_cnt=%d dims=%s", index, block_size, ndim, dims)
    return t, end


class Request:
    """Decodes one request body into ``self.graph`` (reference class of the same name, :18-73)."""

    def __init__(self) -> None:
        self.graph = Graph()

    def decode(self, b: bytes) -> None:
        buf = memoryview(b)
        byte_size, magic, block_cnt, json_size = _HEADER.unpack_from(buf, 0)
        assert magic == REQUEST_MAGIC
        json_end = _HEADER.size + json_size
        json_str = bytes(buf[_HEADER.size:json_end]).decode("utf-8")
        spec = json.loads(json_str)
        pos = align_next(json_end, 4)
        logger.info("decode message: size=%d, json_size=%d, padding=%d, block_cnt=%d",
                    byte_size, json_size, pos - json_end, block_cnt)
        logger.info("json: %s", json_str)

        tensors: List[torch.Tensor] = []
        for i in range(block_cnt):
            t, pos = _read_block(buf, pos, i)
            tensors.append(t)

        g = self.graph
        for node_json in spec["nodes"]:
            g.add_node(node_json["endpoint"], node_json["params"])

        # naming is inverted in the protocol: "in_port" is the SOURCE, "out_port" the DESTINATION
        for edge_json in spec["edges"]:
            dst = edge_json["out_port"]
            tgt_node = g.nodes[dst["node"]]
            if "tensor" in edge_json:
                g.add_input(tensors[edge_json["tensor"]], tgt_node, dst["channel"])
            else:
                src = edge_json["in_port"]
                g.connect(g.nodes[src["node"]], src["channel"], tgt_node, dst["channel"])


def _as_wire_f32(t: torch.Tensor) -> np.ndarray:
    """Row-major float32 host view of a node output.

    The reference requires CPU float32 here (``t.numpy()`` + ``array('f')``, ref :114-115).  Device
    resident or bf16 outputs of the MI355X engine are brought to that form at this single point.
    """
    if t.device.type != "cpu" or t.dtype != torch.float32:
        t = t.detach().to(device="cpu", dtype=torch.float32)
    return np.ascontiguousarray(t.detach().numpy())


_DIRECT_FILL_MIN = 1 << 20


def _bind_pybytes():
    """The two CPython entry points of _writable_bytes, bound ONCE (own function-pointer objects: ctypes.pythonapi's attributes are
    shared process-wide, and re-declaring their signatures on every large response would race with other users of them)."""
    if sys.implementation.name != "cpython":
        return None
    try:
        new = ctypes.PYFUNCTYPE(ctypes.py_object, ctypes.c_void_p, ctypes.c_ssize_t)(("PyBytes_FromStringAndSize", ctypes.pythonapi))
        ptr = ctypes.PYFUNCTYPE(ctypes.c_void_p, ctypes.py_object)(("PyBytes_AsString", ctypes.pythonapi))
        return new, ptr
    except Exception:   # pragma: no cover - exotic builds
        return None


_PYBYTES = _bind_pybytes()


def _writable_bytes(n: int):
    """A new, uninitialised ``bytes`` object of length n and a writable uint8 view of its payload, or None where that is not
    available.  CPython's documented way to build a bytes object in place (PyBytes_FromStringAndSize(NULL, n), then fill the
    buffer PyBytes_AsString returns before anything else sees the object); the view must not outlive the fill."""
    if _PYBYTES is None:
        return None
    try:
        new, ptr = _PYBYTES
        obj = new(None, n)
        view = np.ctypeslib.as_array((ctypes.c_ubyte * n).from_address(ptr(obj)))
        return obj, view
    except Exception:   # pragma: no cover - exotic builds
        return None


class Response:
    """Collects every output of every node and encodes them (reference :76-127)."""

    def __init__(self, graph: Graph) -> None:
        self.outputs: Dict[int, Dict[str, torch.Tensor]] = {}
        for node in graph.nodes:
            for ch, t in node.get_pinout().pinout.items():
                self.set_output(node.index, ch, t)

    def set_output(self, node: int, channel: str, t: torch.Tensor) -> None:
        self.outputs.setdefault(node, {})[channel] = t

    def encode(self) -> bytes:
        labels = []
        tensors: List[torch.Tensor] = []
        shapes: List[Tuple[int, ...]] = []
        for node, outs in self.outputs.items():
            for channel, t in outs.items():
                labels.append({"node": node, "channel": channel})
                tensors.append(t)
                # a lazily synchronised engine output (engine.PendingTensor) knows its shape without waiting for its copy
                shapes.append(tuple(getattr(t, "_ivit_shape", None) or t.shape))

        json_utf8 = json.dumps(labels).encode()
        body_at = align_next(_HEADER.size + len(json_utf8), 4)
        total = body_at + sum(8 + 4 * len(s) + 4 * math.prod(s) for s in shapes)
        if total > 0xFFFFFFFF:
            raise Exception("response exceeds the 4 GiB the u32 byte_size field can describe")
        head = _HEADER.pack(total, RESPONSE_MAGIC, len(tensors), len(json_utf8)) + json_utf8
        if not tensors:
            # reference quirk (ref :108-109): the pad is a seek past the end that nothing follows, so
            # with zero blocks the pad bytes are never materialised although byte_size counts them
            return head
        # One pass over the payload, and each tensor is only brought to the host form - which waits for its device-to-host copy if
        # that is still running - when its turn comes, so node k's bytes are collected while nodes k+1.. still compute.
        # Large responses are written straight into the (not yet shared) bytes object that is returned, so that this copy - 9.7 MB
        # for a ViT-B/16 chain, 0.4 ms with the page faults of a fresh allocation - also runs while the GPU works; otherwise the
        # pieces are joined at the end.  Same bytes either way (no zero-filled staging buffer, no second copy in either).
        fill = _writable_bytes(total) if total >= _DIRECT_FILL_MIN else None
        pieces = [head, b"\0" * (body_at - len(head))]
        pos = 0
        if fill is not None:
            result, view = fill
            for piece in pieces:
                view[pos:pos + len(piece)] = np.frombuffer(piece, dtype=np.uint8)
                pos += len(piece)
        for bi, (shape, t) in enumerate(zip(shapes, tensors)):
            nd = len(shape)
            nbytes = 4 * math.prod(shape)
            block_head = _U32x2.pack(8 + 4 * nd + nbytes, nd) + struct.pack(f"<{nd}I", *shape)
            if fill is None:
                pieces.append(block_head)
                if nbytes:
                    pieces.append(memoryview(_as_wire_f32(t).reshape(-1)).cast("B"))
                continue
            view[pos:pos + len(block_head)] = np.frombuffer(block_head, dtype=np.uint8)
            pos += len(block_head)
            if nbytes:
                data = _as_wire_f32(t).reshape(-1).view(np.uint8)
                if data.size != nbytes:    # the tensor changed shape after its slot was reserved (resize_ / set_ on a node output)
                    raise Exception(f"output '{labels[bi]['channel']}' of node {labels[bi]['node']} changed size while the response was encoded "
                                    f"({data.size} bytes for a block of {nbytes})")
                view[pos:pos + nbytes] = data
                pos += nbytes
        if fill is None:
            return b"".join(pieces)
        assert pos == total
        del view
        return result


def encode_request(nodes: List[dict], edges: List[dict], tensors: List[torch.Tensor]) -> bytes:
    """Client-side request encoder (what ``net_node.js:56-175`` does in the browser).

    Used by tests, bench and smoke to drive the full byte-level path without a browser.
    """
    json_utf8 = json.dumps({"nodes": nodes, "edges": edges}).encode()
    arrays = [_as_wire_f32(t) for t in tensors]
    shapes = [tuple(t.shape) for t in tensors]
    body_at = align_next(_HEADER.size + len(json_utf8), 4)
    total = body_at + sum(8 + 4 * len(s) + 4 * a.size for s, a in zip(shapes, arrays))
    out = bytearray(total)
    _HEADER.pack_into(out, 0, total, REQUEST_MAGIC, len(arrays), len(json_utf8))
    out[_HEADER.size:_HEADER.size + len(json_utf8)] = json_utf8
    pos = body_at
    view = memoryview(out)
    for shape, a in zip(shapes, arrays):
        nd = len(shape)
        nbytes = 4 * a.size
        _U32x2.pack_into(out, pos, 8 + 4 * nd + nbytes, nd)
        struct.pack_into(f"<{nd}I", out, pos + 8, *shape)
        pos += 8 + 4 * nd
        if nbytes:
            view[pos:pos + nbytes] = memoryview(a.reshape(-1)).cast("B")
        pos += nbytes
    return bytes(out)


def decode_response(b: bytes) -> List[Tuple[int, str, torch.Tensor]]:
    """Client-side response decoder (``net_node.js:251-297``): [(node index, channel, tensor)]."""
    buf = memoryview(b)
    byte_size, magic, block_cnt, json_size = _HEADER.unpack_from(buf, 0)
    assert magic == RESPONSE_MAGIC
    labels = json.loads(bytes(buf[_HEADER.size:_HEADER.size + json_size]).decode("utf-8"))
    pos = align_next(_HEADER.size + json_size, 4)
    res = []
    for i in range(block_cnt):
        t, pos = _read_block(buf, pos, i)
        res.append((labels[i]["node"], labels[i]["channel"], t))
    assert byte_size == pos  # the browser asserts this too (net_node.js:294)
    return res
