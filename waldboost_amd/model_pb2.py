"""Runtime-built protobuf classes for the reference wire format.

Mirrors ``waldboost/model.proto`` (reference model.proto:3-23) field for field.
The reference generates ``model_pb2.py`` with protoc (not shipped upstream, and
no protoc in this image), so the descriptor is declared here and the message
classes are obtained from the protobuf runtime.  A ``.pb`` written by either
side parses on the other.
"""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=_F.LABEL_OPTIONAL, type_name=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    return f


def _build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "waldboost_amd/model.proto"
    fd.syntax = "proto3"

    m = fd.message_type.add()
    m.name = "Model"
    _field(m, "shape", 1, _F.TYPE_INT32, _F.LABEL_REPEATED)
    _field(m, "channel_opts", 2, _F.TYPE_MESSAGE, type_name=".ChannelOpts")
    _field(m, "classifier", 3, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".DTree")
    _field(m, "theta", 4, _F.TYPE_FLOAT, _F.LABEL_REPEATED)

    c = fd.message_type.add()
    c.name = "ChannelOpts"
    _field(c, "shrink", 1, _F.TYPE_INT32)
    _field(c, "n_per_oct", 2, _F.TYPE_INT32)
    _field(c, "smooth", 3, _F.TYPE_INT32)
    _field(c, "func", 5, _F.TYPE_STRING)

    t = fd.message_type.add()
    t.name = "DTree"
    _field(t, "feature", 1, _F.TYPE_INT32, _F.LABEL_REPEATED)
    _field(t, "threshold", 2, _F.TYPE_FLOAT, _F.LABEL_REPEATED)
    _field(t, "left", 3, _F.TYPE_INT32, _F.LABEL_REPEATED)
    _field(t, "right", 4, _F.TYPE_INT32, _F.LABEL_REPEATED)
    _field(t, "prediction", 5, _F.TYPE_FLOAT, _F.LABEL_REPEATED)

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = message_factory.GetMessageClass
    return (get(pool.FindMessageTypeByName("Model")),
            get(pool.FindMessageTypeByName("ChannelOpts")),
            get(pool.FindMessageTypeByName("DTree")))


Model, ChannelOpts, DTree = _build()
