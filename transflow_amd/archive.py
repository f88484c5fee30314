"""The on-disk flow format either side of the path: `.flow.zip` archives.

  ZipOutput / NumpyOutput   <- transflow/output/zip.py:6-28, transflow/output/numpy.py:6-14
                               (what Pipeline._setup_flow_export / _update_flow write, pipeline.py:363-377, 505-506)
  ArchiveFlowSource         <- transflow/flow/sources/archive.py:10-51

An archive holds `meta.json` and one `%09d.npy` per frame; frames written by either implementation
are read by the other (tests/test_host_mirror.py).  Host-side I/O only; what is read goes through
FlowSource.post_process on the GPU like any other flow.
"""
from __future__ import annotations

import json
import os
import re
import zipfile

import numpy as np

from .flow import FlowSource


def find_unique_path(path: str) -> str:
    """utils.py:147-160: path, or path with a .NNN counter before the (.flow/.map) extension."""
    root, ext = os.path.splitext(path)
    if root.endswith(".flow") or root.endswith(".map"):
        root, pre_ext = os.path.splitext(root)
        ext = pre_ext + ext
    i = 0
    m = re.match(r".*\.(\d{3})$", root)
    if m:
        i = int(m.group(1)) + 1
        root = root[:-4]
    while os.path.isfile(path):
        path = root + f".{i:03d}" + ext
        i += 1
    return path


class ZipOutput:
    def __init__(self, path: str, replace: bool = False):
        self.path = path if replace else find_unique_path(path)
        if os.path.isfile(self.path):
            os.remove(self.path)
        self.archive = zipfile.ZipFile(self.path, "w", compression=zipfile.ZIP_DEFLATED)

    def write_meta(self, data: dict):
        if not data:
            return
        with self.archive.open("meta.json", "w") as file:
            file.write(json.dumps(data).encode())

    def write_object(self, filename: str, obj: object):
        import pickle
        with self.archive.open(filename, "w") as file:
            pickle.dump(obj, file)

    def close(self):
        self.archive.close()


class NumpyOutput(ZipOutput):
    def __init__(self, path: str, replace: bool = False):
        ZipOutput.__init__(self, path, replace)
        self.index = 0

    def write_array(self, array: np.ndarray):
        with self.archive.open(f"{self.index:09d}.npy", "w") as file:
            np.save(file, array)
        self.index += 1


def flow_export_meta(flow_path, width: int, height: int, framerate, direction, seek_time=None) -> dict:
    """The meta.json of pipeline.py:370-377."""
    return {"path": flow_path, "width": width, "height": height, "framerate": framerate,
            "direction": FlowSource.Direction.from_arg(direction).value, "seek_time": seek_time}


class ArchiveFlowSource(FlowSource):
    class Builder(FlowSource.Builder):
        def __init__(self, path: str, **kwargs):
            super().__init__(**kwargs)
            self.path = path
            self.archive = None

        @property
        def cls(self):
            return ArchiveFlowSource

        def build(self):
            self.archive = zipfile.ZipFile(self.path)
            with self.archive.open("meta.json") as file:
                data = json.loads(file.read().decode())
            # archives without a direction hold forward flows (archive.py:26-27)
            self.direction = FlowSource.Direction(data.get("direction", FlowSource.Direction.FORWARD.value))
            self.width = data["width"]
            self.height = data["height"]
            self.framerate = data["framerate"]
            self.base_length = len(self.archive.infolist()) - 1
            # as in the reference (archive.py:22-31) the base build() is NOT called: no mask / kernel /
            # filters / seek / duration / repeat for archives, `length` stays None, and the source
            # ends with the KeyError of the first missing frame (which the pipeline's source process
            # logs and stops on, pipeline.py:90-97)

        def args(self):
            return [self.archive, *FlowSource.Builder.args(self)]

    def __init__(self, archive: zipfile.ZipFile, *args, **kwargs):
        self.archive = archive
        FlowSource.__init__(self, *args, **kwargs)

    def validate(self):
        super().validate()
        if not isinstance(self.archive, zipfile.ZipFile):
            raise ValueError(f"Attribute archive has incorrect type {type(self.archive)}")

    def next(self):
        with self.archive.open(f"{self.input_frame_index:09d}.npy") as file:
            return np.load(file)

    def close(self):
        self.archive.close()
        super().close()
