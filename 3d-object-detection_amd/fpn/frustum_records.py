"""Frustum record files without TensorFlow: the TFRecord container and the
tf.train.Example messages that dataset/pkl_to_tfrec.py writes (:204-220, :345-367) and
models/model_util.py parses (:89-144), so that real frustums can be fed to get_model.

Container (TFRecord): per record
    uint64 length | uint32 masked_crc32c(length bytes) | data | uint32 masked_crc32c(data)
little-endian, masked_crc = rotate_right(crc32c, 15) + 0xa282ead8 (mod 2^32).

Message (protobuf wire format of tensorflow/core/example/{example,feature}.proto):
    Example  { Features features = 1; }
    Features { map<string, Feature> feature = 1; }       entry: key = 1, value = 2
    Feature  { oneof kind { BytesList bytes_list = 1; FloatList float_list = 2;
                            Int64List int64_list = 3; } }
    BytesList { repeated bytes value = 1; }   FloatList { repeated float value = 1 [packed] }
    Int64List { repeated int64 value = 1 [packed] }
Both packed and unpacked repeated scalars are accepted when reading.

Parsing follows tf.io.parse_single_example with FixedLenFeature specs: a key that is
absent takes the spec's default (error when it has none), a key stored with another list
type than the spec's dtype is an error, and so is a wrong number of values.

KNOWN MISMATCH IN THE REFERENCE (kept, not repaired): the training writer stores
'seg_label' and 'one_hot_vec' as FLOAT lists (pkl_to_tfrec.py:348,355) while the parser
asks for int64 (model_util.py:109,114); the test writer stores 'sample_token' as BYTES
(:210) while the parser asks for int64 (model_util.py:140).  TensorFlow rejects such
records; so does parse_single_example here (ValueError naming the key).  Pass
lenient=True to the parse_* functions to read the writers' files anyway: float lists are
then cast to the requested integer type and a bytes 'sample_token' is returned as bytes.

The reference holds no record file or golden message for this format; the encoder and
decoder are pinned in tests/test_frustum_records.py against the protobuf runtime (an
independent implementation of the wire format) and CRC-32C known answers.
"""
import struct

import numpy as np

NUM_POINT = 1024            # models/model_util.py:103,131
NUM_CHANNELS_OF_PC = 3      # :104,132

# ---------------------------------------------------------------------------
# CRC-32C (Castagnoli), reflected polynomial 0x82F63B78
# ---------------------------------------------------------------------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.arange(256, dtype=np.uint32)
        for _ in range(8):
            t = np.where(t & 1, (t >> 1) ^ np.uint32(0x82F63B78), t >> 1).astype(np.uint32)
        _CRC_TABLE = [int(v) for v in t]
    return _CRC_TABLE


def crc32c(data):
    table = _crc_table()
    crc = 0xFFFFFFFF
    for b in bytes(data):
        crc = table[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    crc = crc32c(data)
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------
# TFRecord container
# ---------------------------------------------------------------------------
def write_records(path, records):
    """Write an iterable of serialized messages; returns the number written."""
    n = 0
    with open(path, "wb") as f:
        for rec in records:
            rec = bytes(rec)
            head = struct.pack("<Q", len(rec))
            f.write(head)
            f.write(struct.pack("<I", masked_crc32c(head)))
            f.write(rec)
            f.write(struct.pack("<I", masked_crc32c(rec)))
            n += 1
    return n


def read_records(path, verify=True):
    """Yield the serialized messages of a record file; with verify, a length or data
    checksum that does not match raises ValueError (TensorFlow: DataLossError)."""
    with open(path, "rb") as f:
        index = 0
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) < 8:
                raise ValueError("record %d: truncated length field" % index)
            crc = f.read(4)
            (length,) = struct.unpack("<Q", head)
            if len(crc) < 4 or (verify and struct.unpack("<I", crc)[0] != masked_crc32c(head)):
                raise ValueError("record %d: corrupted length field" % index)
            data = f.read(length)
            crc = f.read(4)
            if len(data) < length or len(crc) < 4:
                raise ValueError("record %d: truncated record" % index)
            if verify and struct.unpack("<I", crc)[0] != masked_crc32c(data):
                raise ValueError("record %d: corrupted data" % index)
            yield data
            index += 1


# ---------------------------------------------------------------------------
# protobuf wire format, just enough for tf.train.Example
# ---------------------------------------------------------------------------
def _varint(v):
    v &= (1 << 64) - 1          # int64: two's complement, ten bytes when negative
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = 0
    value = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        value |= (b & 0x7F) << shift
        if not b & 0x80:
            return value, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _len_field(number, payload):
    return _varint((number << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf):
    """Yield (field number, wire type, value) of one message; value is an int for
    varint / fixed fields and a memoryview slice for length-delimited ones."""
    buf = memoryview(buf)
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        number, wire = key >> 3, key & 7
        if wire == 0:
            value, pos = _read_varint(buf, pos)
        elif wire == 1:
            value = bytes(buf[pos:pos + 8]); pos += 8
        elif wire == 5:
            value = bytes(buf[pos:pos + 4]); pos += 4
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            if pos + n > len(buf):
                raise ValueError("truncated field %d" % number)
            value = buf[pos:pos + n]; pos += n
        else:
            raise ValueError("unsupported wire type %d" % wire)
        yield number, wire, value


def encode_feature(kind, values):
    """kind 'float' | 'int64' | 'bytes' -> serialized Feature."""
    if kind == "float":
        payload = np.asarray(values, dtype="<f4").ravel().tobytes()
        inner = _len_field(1, payload) if payload else b""
        return _len_field(2, inner)
    if kind == "int64":
        payload = b"".join(_varint(int(v)) for v in np.asarray(values).ravel())
        inner = _len_field(1, payload) if payload else b""
        return _len_field(3, inner)
    if kind == "bytes":
        return _len_field(1, b"".join(_len_field(1, bytes(v)) for v in values))
    raise ValueError("unknown feature kind %r" % (kind,))


def encode_example(features):
    """features: {name: (kind, values)} -> serialized tf.train.Example.  Map entries are
    written in sorted key order (any order is valid; TensorFlow's is unspecified)."""
    entries = b""
    for name in sorted(features):
        kind, values = features[name]
        entry = _len_field(1, name.encode("utf-8")) + _len_field(2, encode_feature(kind, values))
        entries += _len_field(1, entry)
    return _len_field(1, entries)


def _decode_feature(buf):
    kind, out = None, None
    for number, wire, value in _fields(buf):
        if wire != 2:
            continue
        if number == 1:
            kind, out = "bytes", [bytes(v) for n, w, v in _fields(value) if n == 1 and w == 2]
        elif number == 2:
            vals = []
            for n, w, v in _fields(value):
                if n != 1:
                    continue
                vals.append(np.frombuffer(v, dtype="<f4") if w == 2 else np.frombuffer(v, dtype="<f4", count=1))
            kind, out = "float", (np.concatenate(vals) if vals else np.zeros(0, np.float32)).astype(np.float32)
        elif number == 3:
            vals = []
            for n, w, v in _fields(value):
                if n != 1:
                    continue
                if w == 0:
                    vals.append(v)
                else:
                    pos = 0
                    while pos < len(v):
                        x, pos = _read_varint(v, pos)
                        vals.append(x)
            arr = np.array(vals, dtype=np.uint64).astype(np.int64) if vals else np.zeros(0, np.int64)
            kind, out = "int64", arr
    return kind, out


def decode_example(message):
    """serialized tf.train.Example -> {name: (kind, values)}: float32 / int64 arrays or
    a list of bytes; a Feature with no list set decodes to (None, None)."""
    out = {}
    for number, wire, features in _fields(message):
        if number != 1 or wire != 2:
            continue
        for n, w, entry in _fields(features):
            if n != 1 or w != 2:
                continue
            name, feat = None, (None, None)
            for en, ew, ev in _fields(entry):
                if en == 1 and ew == 2:
                    name = bytes(ev).decode("utf-8")
                elif en == 2 and ew == 2:
                    feat = _decode_feature(ev)
            if name is not None:
                out[name] = feat
    return out


# ---------------------------------------------------------------------------
# tf.io.parse_single_example with FixedLenFeature specs
# ---------------------------------------------------------------------------
class FixedLenFeature:
    def __init__(self, shape, dtype, default_value=None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.default_value = default_value


_KIND_OF = {np.dtype(np.float32): "float", np.dtype(np.int64): "int64"}


def parse_single_example(message, spec, lenient=False):
    decoded = decode_example(message)
    out = {}
    for key, f in spec.items():
        want = _KIND_OF[f.dtype]
        size = int(np.prod(f.shape)) if f.shape else 1
        kind, values = decoded.get(key, (None, None))
        if kind is None:
            if f.default_value is None:
                raise ValueError("Feature: %s (data type: %s) is required but could not be found." % (key, want))
            out[key] = np.broadcast_to(np.asarray(f.default_value, dtype=f.dtype), f.shape).copy()
            continue
        if kind != want:
            if not lenient:
                raise ValueError("Key: %s.  Data types don't match. Data type: %s but expected type: %s"
                                 % (key, kind, want))
            if kind == "bytes":
                out[key] = values[0] if (size == 1 and len(values) == 1) else values
                continue
            values = np.asarray(values).astype(f.dtype)
        if len(values) != size:
            raise ValueError("Key: %s.  Number of %s values != expected.  Values size: %d but output shape: %s"
                             % (key, want, len(values), list(f.shape)))
        out[key] = np.asarray(values, dtype=f.dtype).reshape(f.shape)
    return out


def parse_frustum_point_record(tfexample_message, lenient=False):
    """models/model_util.py:101-118 (training / validation records)."""
    num_class = 3
    z = np.zeros
    spec = {
        "size_class": FixedLenFeature((), np.int64, z((), np.int64)),
        "size_residual": FixedLenFeature((3,), np.float32, z((3,), np.float32)),
        "seg_label": FixedLenFeature((NUM_POINT,), np.int64, z((NUM_POINT,), np.int64)),
        "frustum_point_cloud": FixedLenFeature((NUM_POINT, NUM_CHANNELS_OF_PC), np.float32),
        "rot_angle": FixedLenFeature((), np.float32, z((), np.float32)),
        "angle_class": FixedLenFeature((), np.int64, z((), np.int64)),
        "angle_residual": FixedLenFeature((), np.float32, z((), np.float32)),
        "one_hot_vec": FixedLenFeature((num_class,), np.int64),
        "box3d_center": FixedLenFeature((3,), np.float32, z((3,), np.float32)),
    }
    return parse_single_example(tfexample_message, spec, lenient)


def parse_data(raw_record, lenient=False):
    """models/model_util.py:89-98: (point cloud (1024,3) f32, one_hot f32, seg_label i32,
    box3d_center, angle_class i32, angle_residual, size_class i32, size_residual)."""
    e = parse_frustum_point_record(raw_record, lenient)
    return (e["frustum_point_cloud"], e["one_hot_vec"].astype(np.float32), e["seg_label"].astype(np.int32),
            e["box3d_center"], e["angle_class"].astype(np.int32), e["angle_residual"],
            e["size_class"].astype(np.int32), e["size_residual"])


def parse_frustum_point_test_record(tfexample_message, lenient=False):
    """models/model_util.py:129-144 (detection-time records)."""
    z = np.zeros
    spec = {
        "frustum_point_cloud": FixedLenFeature((NUM_POINT, NUM_CHANNELS_OF_PC), np.float32),
        "rot_angle": FixedLenFeature((), np.float32, z((), np.float32)),
        "one_hot_vec": FixedLenFeature((3,), np.int64),
        "prob": FixedLenFeature((), np.float32, z((), np.float32)),
        "type_name": FixedLenFeature((), np.int64),
        "sample_token": FixedLenFeature((), np.int64),
        "box_2d": FixedLenFeature((4,), np.float32),
    }
    return parse_single_example(tfexample_message, spec, lenient)


def parse_test_data(raw_record, lenient=False):
    """models/model_util.py:121-126: (point cloud, one_hot f32, rot_angle, prob,
    type_name, sample_token, box_2d)."""
    e = parse_frustum_point_test_record(raw_record, lenient)
    return (e["frustum_point_cloud"], e["one_hot_vec"].astype(np.float32), np.float32(e["rot_angle"]),
            np.float32(e["prob"]), e["type_name"], e["sample_token"], e["box_2d"])


# ---------------------------------------------------------------------------
# the writers' messages (dataset/pkl_to_tfrec.py)
# ---------------------------------------------------------------------------
def serialize_train_example(point_set, seg, box3d_center, angle_class, angle_residual, size_class, size_residual,
                            rot_angle, one_hot_vec):
    """tfrec_Gen_Train_Val.serialize_example (:345-357), list types as the reference
    writes them (seg_label and one_hot_vec as float lists)."""
    return encode_example({
        "frustum_point_cloud": ("float", np.ravel(point_set)),
        "seg_label": ("float", seg),
        "box3d_center": ("float", box3d_center),
        "angle_class": ("int64", [angle_class]),
        "angle_residual": ("float", [angle_residual]),
        "size_class": ("int64", [size_class]),
        "size_residual": ("float", np.ravel(size_residual)),
        "rot_angle": ("float", [rot_angle]),
        "one_hot_vec": ("float", one_hot_vec),
    })


def serialize_test_example(point_set, rot_angle, prob_value, one_hot_vec, cls_index, scene_id, box_2d):
    """tfrecGen_test.serialize_example (:204-213)."""
    return encode_example({
        "frustum_point_cloud": ("float", np.ravel(point_set)),
        "rot_angle": ("float", [rot_angle]),
        "one_hot_vec": ("int64", one_hot_vec),
        "prob": ("float", [prob_value]),
        "type_name": ("int64", [cls_index]),
        "sample_token": ("bytes", [scene_id.encode("utf-8")]),
        "box_2d": ("float", np.ravel(box_2d)),
    })


def frustum_batches(path, batch_size, device, test=False, lenient=True, drop_remainder=False):
    """Feed a record file to get_model: yields (point_cloud (B,1024,3) float32, one_hot
    (B,3) float32, rest) with the two tensors on `device` and `rest` the list of the
    records' remaining fields.  Host buffers are pinned and the copies are queued on the
    current stream, so the next batch's parsing overlaps the device work of this one."""
    import torch
    parse = parse_test_data if test else parse_data
    pcs, ohs, rest = [], [], []

    pin = torch.device(device).type == "cuda"

    def flush():
        pc = torch.from_numpy(np.stack(pcs))
        oh = torch.from_numpy(np.stack(ohs))
        if pin:
            pc, oh = pc.pin_memory(), oh.pin_memory()
        out = (pc.to(device, non_blocking=True), oh.to(device, non_blocking=True), list(rest))
        pcs.clear(); ohs.clear(); rest.clear()
        return out
    for rec in read_records(path):
        fields = parse(rec, lenient)
        pcs.append(fields[0]); ohs.append(fields[1]); rest.append(fields[2:])
        if len(pcs) == batch_size:
            yield flush()
    if pcs and not drop_remainder:
        yield flush()
