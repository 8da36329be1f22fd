"""volym_amd.devtools (volym_devtools/src/main.rs:15-95): .seg.nrrd -> segments.json + label bytes, and the JSON it
writes feeds the importance map the way the shipped assets/boston_teapot_*_segments.json does."""
import json
import os

import numpy as np


def test_seg_nrrd_roundtrip(tmp_path, volym_lib):
    from volym_amd import __main__ as cli, scene
    labels = (np.arange(6 * 5 * 4) % 5).astype(np.uint8)
    labels[7] = 10                                     # a newline byte inside the payload
    header = (b"NRRD0004\n# a 3D Slicer segmentation\ntype: unsigned char\ndimension: 3\nsizes: 6 5 4\nencoding: raw\n"
              b"Segment0_ID:=Segment_1\nSegment0_LabelValue:=2\nSegment0_Name:=Lobster\n"
              b"Segment1_Name:=Cup\nSegment1_ID:=Segment_2\nSegment1_LabelValue:=3\n"
              b"Segment2_LabelValue:=4\nSegment2_Name:=Ground\nSegment2_ID:=Segment_3\n\n")
    nrrd = tmp_path / "t.seg.nrrd"
    nrrd.write_bytes(header + labels.tobytes())
    js, raw = str(tmp_path / "segs.json"), str(tmp_path / "segs.raw")
    assert cli.main(["devtools", str(nrrd), js, raw]) == 0
    segs = json.load(open(js))
    assert [(s["index"], s["name"], s["id"], s["label_value"], s["importance"]) for s in segs] == [
        (0, "Lobster", "Segment_1", 2, 0), (1, "Cup", "Segment_2", 3, 0), (2, "Ground", "Segment_3", 4, 0)]
    assert np.array_equal(np.fromfile(raw, np.uint8), labels)
    # same schema as the reference's shipped JSON: usable as is by the importance upload
    shipped = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "boston_teapot_segments.json")))
    assert set(shipped[0].keys()) == set(segs[0].keys())
    segs[0]["importance"] = 255
    imp = scene.map_segments_to_importance(labels, scene.load_segments(segs))
    assert set(np.unique(imp)) == {0, 255} and np.array_equal(imp == 255, labels == 2)


def test_segment_index_above_255_is_refused():
    """volym_devtools/src/main.rs:50 parses the index as a u8 and panics above 255; a silent wrap would overwrite Segment0"""
    import pytest
    from volym_amd import devtools
    lines = [b"Segment0_Name:=A", b"Segment0_ID:=a", b"Segment0_LabelValue:=1", b"Segment256_Name:=B"]
    with pytest.raises(ValueError):
        devtools.read_segments(lines)
