// ot_writer — writes a LibTorch archive exactly as tch's VarStore::save does (network/mod.rs:16-18 -> torch-sys
// at_save_multi: OutputArchive::write(name, tensor) for every variable, then save_to), so that the reference's
// Net::load reads models trained here.  Host tool of takzero_amd.ot.save_ot; input: a text manifest of
// "<archive name> <n dims> <dims...> <raw f32 file>" lines.  Built on demand with g++ against the torch wheel.
#include <torch/torch.h>

#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    std::ifstream manifest(argv[1]);
    torch::serialize::OutputArchive archive;
    std::string line;
    while (std::getline(manifest, line)) {
        std::istringstream ls(line);
        std::string name, file;
        int nd;
        ls >> name >> nd;
        std::vector<int64_t> dims(nd);
        int64_t total = 1;
        for (auto& d : dims) {
            ls >> d;
            total *= d;
        }
        ls >> file;
        std::vector<float> buf(total);
        std::ifstream f(file, std::ios::binary);
        f.read(reinterpret_cast<char*>(buf.data()), total * sizeof(float));
        if (!f) return 3;
        archive.write(name, torch::from_blob(buf.data(), dims, torch::kFloat32).clone(), /*is_buffer=*/false);
    }
    archive.save_to(argv[2]);
    return 0;
}
