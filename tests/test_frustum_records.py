"""Input side of the path: TFRecord container, tf.train.Example messages, the reference's
record parsers (models/model_util.py:89-144), the pickle-list dataset
(scripts/provider.py:107-275) and the record writers (dataset/pkl_to_tfrec.py).

The reference holds no record or pickle file, and TensorFlow is absent, so the wire format
is pinned against independent implementations: CRC-32C known answers (RFC 3720) and the
protobuf runtime with the Example / Feature schema declared here; the dataset logic is
pinned against the formulas of the reference re-evaluated by hand in the tests."""
import os
import pickle

import numpy as np
import pytest

from fpn import frustum_records as R
from fpn import pkl_to_tfrec as P
from fpn import provider


def test_crc32c_known_answers():
    assert R.crc32c(b"123456789") == 0xE3069283
    assert R.crc32c(bytes(32)) == 0x8A9136AA
    assert R.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert R.crc32c(bytes(range(32))) == 0x46DD794E
    # TFRecord mask of a known crc
    crc = 0xE3069283
    assert R.masked_crc32c(b"123456789") == ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_record_container_round_trip_and_corruption(tmp_path):
    path = str(tmp_path / "a.tfrec")
    recs = [b"", b"x", os.urandom(1000), b"tail"]
    assert R.write_records(path, recs) == 4
    assert list(R.read_records(path)) == recs
    raw = bytearray(open(path, "rb").read())
    raw[12 + 4 + 8 + 4 + 0] ^= 1            # first byte of the second record's data
    open(path, "wb").write(raw)
    with pytest.raises(ValueError, match="record 1: corrupted data"):
        list(R.read_records(path))
    assert len(list(R.read_records(path, verify=False))) == 4
    open(path, "wb").write(raw[:-3])
    with pytest.raises(ValueError, match="truncated"):
        list(R.read_records(path, verify=False))


def example_classes():
    """tf.train.Example and friends declared to the protobuf runtime (field numbers of
    tensorflow/core/example/feature.proto and example.proto)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto()
    f.name = "fpn_test_example.proto"
    f.package = "fpn_test"
    f.syntax = "proto3"
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = f.message_type.add()
        m.name = name
        return m

    def field(m, name, number, typ, label=T.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
        x = m.field.add()
        x.name, x.number, x.type, x.label = name, number, typ, label
        if type_name:
            x.type_name = type_name
        if packed is not None:
            x.options.packed = packed
        if oneof is not None:
            x.oneof_index = oneof
        return x
    field(msg("BytesList"), "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    field(msg("FloatList"), "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    field(msg("Int64List"), "value", 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    feat = msg("Feature")
    feat.oneof_decl.add().name = "kind"
    field(feat, "bytes_list", 1, T.TYPE_MESSAGE, type_name=".fpn_test.BytesList", oneof=0)
    field(feat, "float_list", 2, T.TYPE_MESSAGE, type_name=".fpn_test.FloatList", oneof=0)
    field(feat, "int64_list", 3, T.TYPE_MESSAGE, type_name=".fpn_test.Int64List", oneof=0)
    feats = msg("Features")
    entry = feats.nested_type.add()
    entry.name = "FeatureEntry"
    entry.options.map_entry = True
    field(entry, "key", 1, T.TYPE_STRING)
    field(entry, "value", 2, T.TYPE_MESSAGE, type_name=".fpn_test.Feature")
    field(feats, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".fpn_test.Features.FeatureEntry")
    field(msg("Example"), "features", 1, T.TYPE_MESSAGE, type_name=".fpn_test.Features")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("fpn_test.Example"))


def test_example_messages_against_protobuf_runtime():
    Example = example_classes()
    rng = np.random.RandomState(0)
    feats = {
        "frustum_point_cloud": ("float", rng.standard_normal(3072).astype(np.float32)),
        "angle_class": ("int64", [7]),
        "negative": ("int64", [-1, -(2 ** 40), 2 ** 62, 0, 300]),
        "sample_token": ("bytes", [b"abc\x00def", b""]),
        "empty_float": ("float", []),
    }
    mine = R.encode_example(feats)
    # the runtime reads what this encoder wrote ...
    ex = Example()
    ex.ParseFromString(mine)
    fm = ex.features.feature
    np.testing.assert_array_equal(np.array(fm["frustum_point_cloud"].float_list.value, np.float32),
                                  feats["frustum_point_cloud"][1])
    assert list(fm["angle_class"].int64_list.value) == [7]
    assert list(fm["negative"].int64_list.value) == feats["negative"][1]
    assert list(fm["sample_token"].bytes_list.value) == feats["sample_token"][1]
    assert len(fm["empty_float"].float_list.value) == 0
    # ... and this decoder reads what the runtime wrote
    ex2 = Example()
    ex2.features.feature["a"].float_list.value.extend([1.5, -2.25, 3.0])
    ex2.features.feature["b"].int64_list.value.extend([-5, 1 << 50])
    ex2.features.feature["c"].bytes_list.value.extend([b"tok"])
    got = R.decode_example(ex2.SerializeToString())
    assert got["a"][0] == "float" and got["a"][1].tolist() == [1.5, -2.25, 3.0]
    assert got["b"][0] == "int64" and got["b"][1].tolist() == [-5, 1 << 50]
    assert got["c"] == ("bytes", [b"tok"])
    back = R.decode_example(mine)
    assert back["negative"][1].tolist() == feats["negative"][1]
    assert back["sample_token"][1] == feats["sample_token"][1]


def frustum_lists(n_items=5, seed=0, lyft=False, rgb=False):
    rng = np.random.RandomState(seed)
    types = ['car', 'pedestrian', 'bicycle'] if lyft else ['Car', 'Pedestrian', 'Cyclist']
    ids = ["tok%d" % i for i in range(n_items)]
    box2d = [rng.uniform(0, 100, 4) for _ in range(n_items)]
    pts = [rng.standard_normal((rng.choice([300, 1500]), 4)) * [2, 1, 2, 1] + [0, 0, 20, 0] for _ in range(n_items)]
    typ = [types[i % 3] for i in range(n_items)]
    ang = [float(rng.uniform(-np.pi, np.pi)) for _ in range(n_items)]
    if rgb:
        return [ids, box2d, pts, typ, ang, [float(p) for p in rng.uniform(0, 1, n_items)]]
    box3d = [rng.standard_normal((8, 3)) + [1, 1, 20] for _ in range(n_items)]
    labels = [(rng.uniform(0, 1, p.shape[0]) < 0.3).astype(np.float64) for p in pts]
    heading = [float(rng.uniform(-np.pi, np.pi)) for _ in range(n_items)]
    size = [rng.uniform(1, 5, 3) for _ in range(n_items)]
    return [ids, box2d, box3d, pts, labels, typ, heading, size, ang]


def test_frustum_dataset_items_follow_the_reference_formulas(tmp_path):
    lists = frustum_lists()
    path = str(tmp_path / "f.pickle")
    with open(path, "wb") as fp:
        for l in lists:
            pickle.dump(l, fp)
    ds = provider.FrustumDataset(1024, "val", rotate_to_center=True, overwritten_data_path=path, one_hot=True)
    assert len(ds) == 5
    for index in range(5):
        np.random.seed(100 + index)
        pc, seg, center, acls, ares, scls, sres, rot, one_hot = ds[index]
        # the same draws by hand
        np.random.seed(100 + index)
        src = lists[3][index]
        n = src.shape[0]
        choice = np.random.choice(n, 1024, replace=n < 1024)
        assert rot == np.pi / 2 + lists[8][index]
        c, s = np.cos(rot), np.sin(rot)
        want = src[choice].copy()
        x, z = want[:, 0].copy(), want[:, 2].copy()
        want[:, 0] = c * x - s * z            # [x z] . transpose([[c,-s],[s,c]])
        want[:, 2] = s * x + c * z
        np.testing.assert_allclose(pc, want, rtol=0, atol=1e-12)
        assert pc.shape == (1024, 4)
        np.testing.assert_array_equal(seg, lists[4][index][choice])
        if n >= 1024:
            assert len(set(choice.tolist())) == 1024
        b = lists[2][index]
        ctr = (b[0] + b[6]) / 2
        np.testing.assert_allclose(center, [c * ctr[0] - s * ctr[2], ctr[1], s * ctr[0] + c * ctr[2]], atol=1e-12)
        k, r = provider.angle2class(lists[6][index] - rot, 12)
        assert (acls, ares) == (k, r)
        assert abs((acls * (2 * np.pi / 12) + ares - (lists[6][index] - rot)) % (2 * np.pi)) < 1e-9 or \
            abs((acls * (2 * np.pi / 12) + ares - (lists[6][index] - rot)) % (2 * np.pi) - 2 * np.pi) < 1e-9
        assert scls == provider.g_type2class[lists[5][index]]
        np.testing.assert_array_equal(sres, lists[7][index] - provider.g_type_mean_size[lists[5][index]])
        assert one_hot.tolist() == [float(j == index % 3) for j in range(3)]
    # augmentation draws come after the resampling choice, flip before shift
    aug = provider.FrustumDataset(1024, "train", random_flip=True, random_shift=True, rotate_to_center=True,
                                  lists=lists, one_hot=False)
    np.random.seed(7)
    pc, seg, center, *_ = aug[1]
    np.random.seed(7)
    n = lists[3][1].shape[0]
    np.random.choice(n, 1024, replace=n < 1024)
    flip = np.random.random() > 0.5
    base = ds.get_center_view_box3d_center(1)
    cx = -base[0] if flip else base[0]
    dist = np.sqrt(cx ** 2 + base[1] ** 2)
    shift = np.clip(np.random.randn() * dist * 0.05, dist * 0.8, dist * 1.2)
    np.testing.assert_allclose(center, [cx, base[1], base[2] + shift], atol=1e-12)
    # detection-time items
    rgb = provider.FrustumDataset(512, "val", rotate_to_center=True, from_rgb_detection=True, one_hot=True,
                                  lists=frustum_lists(rgb=True))
    item = rgb[0]
    assert len(item) == 4 and item[0].shape == (512, 4)


def test_frustum_pickle_reader_builds_nothing_but_lists_and_arrays(tmp_path):
    path = str(tmp_path / "evil.pickle")

    class Evil(object):
        def __reduce__(self):
            return (os.getcwd, ())
    with open(path, "wb") as fp:
        pickle.dump([Evil()], fp)
    with pytest.raises(pickle.UnpicklingError, match="only lists"):
        provider.load_frustum_lists(path, 1)


def test_writers_and_parsers_round_trip(tmp_path):
    gen = P.tfrec_Gen_Train_Val(lists=frustum_lists(lyft=True, seed=3))
    path = str(tmp_path / "train.tfrec")
    np.random.seed(11)
    assert gen.write_tfrec(path) == 5
    np.random.seed(11)
    want = [gen.feature_extraction(i) for i in range(5)]
    recs = list(R.read_records(path))
    # the reference's own writer/parser disagreement: float lists where int64 is asked for
    with pytest.raises(ValueError, match="Key: (seg_label|one_hot_vec).  Data types don't match"):
        R.parse_data(recs[0])
    for rec, w in zip(recs, want):
        pc, one_hot, seg, center, acls, ares, scls, sres = R.parse_data(rec, lenient=True)
        assert pc.dtype == np.float32 and pc.shape == (1024, 3) and seg.dtype == np.int32 and one_hot.dtype == np.float32
        np.testing.assert_array_equal(pc, w[0].astype(np.float32))
        np.testing.assert_array_equal(seg, w[1].astype(np.int32))
        np.testing.assert_array_equal(center, w[2].astype(np.float32))
        assert (int(acls), int(scls)) == (w[3], w[5])
        assert ares == np.float32(w[4])
        np.testing.assert_array_equal(sres, w[6].astype(np.float32))
        np.testing.assert_array_equal(one_hot, w[8].astype(np.float32))
    # defaults for absent keys, errors for absent required keys and wrong sizes
    msg = R.encode_example({"frustum_point_cloud": ("float", np.zeros(3072)), "one_hot_vec": ("int64", [0, 1, 0])})
    e = R.parse_frustum_point_record(msg)
    assert e["size_class"] == 0 and e["seg_label"].shape == (1024,) and e["box3d_center"].tolist() == [0, 0, 0]
    with pytest.raises(ValueError, match="frustum_point_cloud.*required"):
        R.parse_frustum_point_record(R.encode_example({"one_hot_vec": ("int64", [0, 1, 0])}))
    with pytest.raises(ValueError, match="Number of float values"):
        R.parse_frustum_point_record(R.encode_example({"frustum_point_cloud": ("float", np.zeros(10)),
                                                       "one_hot_vec": ("int64", [0, 1, 0])}))

    tgen = P.tfrecGen_test(lists=frustum_lists(lyft=True, rgb=True, seed=4))
    tpath = str(tmp_path / "test.tfrec")
    np.random.seed(5)
    assert tgen.write_tfrec(tpath) == 5
    np.random.seed(5)
    twant = [tgen.feature_extraction(i) for i in range(5)]
    trecs = list(R.read_records(tpath))
    with pytest.raises(ValueError, match="Key: sample_token"):
        R.parse_test_data(trecs[0])
    for rec, w in zip(trecs, twant):
        pc, one_hot, rot, prob, type_name, token, box2d = R.parse_test_data(rec, lenient=True)
        np.testing.assert_array_equal(pc, w[0].astype(np.float32))
        assert rot == np.float32(w[1]) and prob == np.float32(w[2]) and int(type_name) == w[4]
        assert token == w[5].encode() and one_hot.tolist() == w[3].astype(np.float32).tolist()
        np.testing.assert_array_equal(box2d, w[6].astype(np.float32))
    # batches for get_model (host tensors here; on a GPU box the same call pins and copies)
    batches = list(R.frustum_batches(tpath, 2, "cpu", test=True))
    assert [b[0].shape[0] for b in batches] == [2, 2, 1] and batches[0][0].shape == (2, 1024, 3)
    assert batches[0][1].shape == (2, 3) and len(batches[0][2]) == 2
