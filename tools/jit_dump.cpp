// Dumps the HIP source of the specialised constraint kernels of a DAG blob (no GPU needed):
//   MH_JIT=1 MH_JIT_NO_COMPILE=1 MH_JIT_DUMP=<dir> tools/jit_dump <blob.bin>
#include "../miden-vm_amd/csrc/air.hpp"
#include <cstdio>
#include <fstream>
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  try {
    DagIR ir = dag_parse((const u64*)raw.data(), raw.size() / 8);
    size_t gates = 0;
    for (size_t i = 0; i < ir.nodes.size(); i++) gates += ir.live[i] && dag_is_gate(ir.nodes[i].op);
    printf("nodes %zu, live gates %zu, constraints %zu\n", ir.nodes.size(), gates, ir.cons.size());
    jit_program_build(nullptr, ir);
  } catch (const std::exception& e) {
    printf("error: %s\n", e.what());
    return 1;
  }
  return 0;
}
