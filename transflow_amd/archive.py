"""`.flow.zip` flow archives -- the on-disk flow format either side of the path.

The format, as the reference's files lay it out (written by transflow/output/zip.py:6-28 and
output/numpy.py:6-14 from pipeline.py:363-377, 505-506; read by flow/sources/archive.py:10-51): a
deflated zip with one member `meta.json` ({"path", "width", "height", "framerate", "direction",
"seek_time"}) followed by one `.npy` member per frame named by its nine-digit index.  This module is
an independent reader / writer of that layout: `FlowArchiveWriter`, `read_archive_meta`,
`read_archive_frame`, and `ArchiveFlowSource`, whose frames go through FlowSource.post_process on the
GPU like any other flow.  tests/test_host_mirror.py checks that archives written here are
byte-identical to the reference's and that each implementation reads the other's.
"""
from __future__ import annotations

import io
import json
import os
import re
import zipfile

import numpy as np

from .flow import FlowSource

META_MEMBER = "meta.json"


def frame_member(index: int) -> str:
    return "%09d.npy" % index


def unique_path(path: str) -> str:
    """`path` if nothing is there yet, otherwise the first free `<stem>.NNN<ext>`; a `.flow` / `.map`
    before the extension belongs to the extension (`a.flow.zip` -> `a.000.flow.zip`), and a stem that
    already ends in a counter continues from it (utils.find_unique_path's naming, utils.py:147-160)."""
    if not os.path.isfile(path):
        return path
    stem, ext = os.path.splitext(path)
    for tag in (".flow", ".map"):
        if stem.endswith(tag):
            stem, ext = stem[:-len(tag)], tag + ext
            break
    counter = re.search(r"\.(\d{3})$", stem)
    n = 0
    if counter:
        stem, n = stem[:counter.start()], int(counter.group(1)) + 1
    while True:
        candidate = f"{stem}.{n:03d}{ext}"
        if not os.path.isfile(candidate):
            return candidate
        n += 1


def flow_export_meta(flow_path, width: int, height: int, framerate, direction, seek_time=None) -> dict:
    """The six fields pipeline.py:370-377 stores with an exported flow."""
    return dict(path=flow_path, width=width, height=height, framerate=framerate,
                direction=FlowSource.Direction.from_arg(direction).value, seek_time=seek_time)


class FlowArchiveWriter:
    """Appends frames to a new archive.  `replace=False` never overwrites: it picks `unique_path(path)`."""

    def __init__(self, path: str, replace: bool = False):
        self.path = path if replace else unique_path(path)
        self._zip = zipfile.ZipFile(self.path, mode="w", compression=zipfile.ZIP_DEFLATED)
        self.index = 0

    def _put(self, member: str, payload: bytes) -> None:
        with self._zip.open(member, mode="w") as f:
            f.write(payload)

    def write_meta(self, meta: dict) -> None:
        if meta:
            self._put(META_MEMBER, json.dumps(meta).encode())

    def write_array(self, array: np.ndarray) -> None:
        buf = io.BytesIO()
        np.save(buf, array)
        self._put(frame_member(self.index), buf.getvalue())
        self.index += 1

    def close(self) -> None:
        self._zip.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


NumpyOutput = FlowArchiveWriter      # the name the reference's pipeline uses for this role (output/numpy.py)


def read_archive_meta(zf: zipfile.ZipFile) -> dict:
    return json.loads(zf.read(META_MEMBER).decode())


def read_archive_frame(zf: zipfile.ZipFile, index: int) -> np.ndarray:
    """Frame `index`; a missing member is zipfile's KeyError -- that is how an archive ends."""
    return np.load(io.BytesIO(zf.read(frame_member(index))))


class ArchiveFlowSource(FlowSource):
    """Flows replayed from an archive.  Like the reference's (archive.py:22-31) its builder takes
    geometry, frame rate and direction from meta.json and does no timing arithmetic at all: no seek,
    duration or repeat for archives, `length` stays None, and iteration ends with the KeyError of the
    first missing frame (which the pipeline's source process logs and stops on, pipeline.py:90-97)."""

    class Builder(FlowSource.Builder):
        def __init__(self, path: str, **kwargs):
            FlowSource.Builder.__init__(self, **kwargs)
            self.path, self.archive = path, None

        cls = property(lambda self: ArchiveFlowSource)

        def build(self):
            self.archive = zipfile.ZipFile(self.path)
            meta = read_archive_meta(self.archive)
            self.width, self.height, self.framerate = meta["width"], meta["height"], meta["framerate"]
            # archives from before the field existed hold forward flows (archive.py:26-27)
            self.direction = FlowSource.Direction(meta.get("direction", FlowSource.Direction.FORWARD.value))
            self.base_length = len(self.archive.namelist()) - 1

        def args(self):
            return [self.archive] + FlowSource.Builder.args(self)

    def __init__(self, archive: zipfile.ZipFile, *args, **kwargs):
        self.archive = archive
        FlowSource.__init__(self, *args, **kwargs)

    def validate(self):
        FlowSource.validate(self)
        if not isinstance(self.archive, zipfile.ZipFile):
            raise ValueError(f"Attribute archive has incorrect type {type(self.archive)}")

    def next(self):
        return read_archive_frame(self.archive, self.input_frame_index)

    def close(self):
        self.archive.close()
        FlowSource.close(self)
