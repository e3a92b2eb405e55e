// config_dump.cpp -- loads run configurations with shim/micv_config.hpp and prints every leaf
// (Node::dump), so that tests/test_config.py can compare the C++ reader with the Python one on the
// reference's own files.  `--ps7` adds the typed ps7 views (Config::MHI, loadActionLengths).
#include <cstdio>
#include <cstring>
#include <iostream>

#include "../../introtocomputervision_amd/shim/micv_config.hpp"

int main(int argc, char **argv) {
    try {
        for (int i = 1; i < argc; i++) {
            if (!std::strcmp(argv[i], "--ps7")) {
                const micv_config::Node cfg = micv_config::Node::load(argv[++i]);
                for (auto &kv : micv_config::last_frames(cfg)) std::cout << kv.first << " " << kv.second << "\n";
                const micv_config::MHI m(cfg.child("mhi_action3"));
                std::cout << "mhi_action3 " << m.diff_threshold << " " << m.pre_blur_size << " " << m.pre_blur_sigma << " " << m.tau << "\n";
                continue;
            }
            micv_config::Node::load(argv[i]).dump(std::cout);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
