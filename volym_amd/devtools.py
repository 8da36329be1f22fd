"""`volym_devtools in.seg.nrrd segments.json segments.raw` (volym_devtools/src/main.rs:15-95): the offline step that turns a
3D-Slicer segmentation into the two files the importance upload reads (src/demos/simple/importance.rs:13-20, :45-60).

* segments.json: for every `Segment<N>_Name:=`, `Segment<N>_ID:=`, `Segment<N>_LabelValue:=` triple of the NRRD header one
  {"index", "name", "id", "label_value", "importance": 0} (main.rs:35-82; the reference iterates a HashMap, i.e. in no
  particular order -- here sorted by index); the user then edits the importances.
* segments.raw: the label bytes.  The reference writes "the last text line of the file" (main.rs:85-95), which is the
  payload only as long as the payload holds no newline byte (and its line reader rejects bytes that are not UTF-8); here
  the payload is what NRRD says it is: everything after the first blank line of a `raw`-encoded file.
"""
import json
import re

_NAME = re.compile(rb"Segment(\d+)_Name:=(.*)")
_ID = re.compile(rb"Segment(\d+)_ID:=(.*)")
_LABEL = re.compile(rb"Segment(\d+)_LabelValue:=(.*)")


def split_nrrd(data):
    """bytes of a .nrrd -> (header lines, payload bytes)"""
    for sep in (b"\r\n\r\n", b"\n\n"):
        i = data.find(sep)
        if i >= 0:
            return data[:i].replace(b"\r\n", b"\n").split(b"\n"), data[i + len(sep):]
    return data.replace(b"\r\n", b"\n").split(b"\n"), b""


def _segment_index(m):
    """Segment<N>: the reference parses N as a u8 and panics above 255 (volym_devtools/src/main.rs:50)"""
    idx = int(m.group(1))
    if idx > 255:
        raise ValueError("segment index %d does not fit a u8 (volym_devtools/src/main.rs:50 refuses it)" % idx)
    return idx


def read_segments(header_lines):
    names, ids, labels = {}, {}, {}
    for line in header_lines:
        m = _NAME.search(line)
        if m:
            names[_segment_index(m)] = m.group(2).decode("utf-8", "replace")
            continue
        m = _ID.search(line)
        if m:
            ids[_segment_index(m)] = m.group(2).decode("utf-8", "replace")
            continue
        m = _LABEL.search(line)
        if m:
            labels[_segment_index(m)] = int(m.group(2))
    segs = []
    for index in sorted(names):
        if index not in ids or index not in labels:
            raise ValueError("Segment%d has a name but no ID / LabelValue" % index)      # the reference unwrap()s here
        if not 0 <= labels[index] <= 255:
            raise ValueError("Segment%d_LabelValue does not fit a u8" % index)
        segs.append({"index": index, "name": names[index], "id": ids[index], "label_value": labels[index], "importance": 0})
    return segs


def convert(nrrd_path, json_path, raw_path):
    data = open(nrrd_path, "rb").read()
    header, payload = split_nrrd(data)
    enc = [l.split(b":", 1)[1].strip().lower() for l in header if l.lower().startswith(b"encoding:")]
    if enc and enc[0] != b"raw":
        raise ValueError("only raw-encoded NRRD files are supported (encoding: %s)" % enc[0].decode())
    segs = read_segments(header)
    with open(json_path, "w") as f:
        json.dump(segs, f)
    with open(raw_path, "wb") as f:
        f.write(payload)
    return segs, len(payload)
