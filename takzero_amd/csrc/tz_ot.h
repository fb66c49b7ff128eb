// tz_ot.h — the reference's model files without LibTorch (csrc/tz_ot.cpp).
//
// `learn` writes model_latest.ot / model_NNNNNNN.ot with tch's VarStore::save (takzero/src/network/mod.rs:16-18), i.e.
// torch-sys at_save_multi: torch::serialize::OutputArchive::write(name, tensor) per variable + save_to = a TorchScript
// module archive: a zip (stored entries, 64-byte aligned payloads) with `<stem>/data.pkl` (a protocol-2 pickle of the module
// object: attribute name -> torch._utils._rebuild_tensor_v2(storage, offset, size, stride, ...)), `<stem>/data/<k>` (raw
// little-endian storages), `<stem>/code/__torch__.py`, `<stem>/constants.pkl`, `<stem>/version`.  tch / LibTorch are not
// under /root/reference (registry crates tch 0.22.0 / torch-sys 0.22.0, Cargo.lock:1596-1598); the format is pinned here by
// archives that the LibTorch of this image writes and reads (tests/test_ot_native.py).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

struct HostTensor {
    std::vector<uint32_t> dims;
    std::vector<float> data;
};
// variables by name, in the order they were created / read
typedef std::vector<std::pair<std::string, HostTensor>> NamedTensors;
typedef std::map<std::string, HostTensor> TensorStore;

// LibTorch archive -> tensors under the names stored in it (tch VarStore names).  Returns TZ_OK or TZ_EPARSE.
int ot_read_archive(const unsigned char* data, size_t bytes, NamedTensors& out);
// tensors -> LibTorch archive bytes, byte-compatible with OutputArchive::save_to (stem = archive directory name)
int ot_write_archive(const NamedTensors& tensors, const std::string& stem, std::vector<unsigned char>& out);
// tch VarStore names -> the `.a.` / `.b.` spelling used inside this library: both SmallBlocks of a ResidualBlock are created
// under one path (residual.rs:50-55), tch renames the second one's variables `<name>__<K>`
int ot_canonical_names(const NamedTensors& in, TensorStore& out);
// the inverse: creation order of net5.rs:44-148 / net6_simhash.rs:43-141, K = variables registered so far
void ot_tch_names(const TensorStore& in, NamedTensors& out);
// the flat .tzw container (takzero_amd/weights.py)
int tzw_parse(const unsigned char* data, size_t bytes, TensorStore& out);
void tzw_dump(const TensorStore& in, std::vector<unsigned char>& out);
// file helpers: reads .ot (zip magic) or .tzw (TZW1 magic) by content
int weights_read_file(const char* path, TensorStore& out);
// writes by extension (.tzw, anything else = LibTorch archive) to `path`.part and renames: a reader never sees half a model
int weights_write_file(const char* path, const TensorStore& in);
