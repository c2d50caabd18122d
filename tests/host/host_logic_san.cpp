// Host-only unit test of the C-ABI layer's CPU code paths (constant folding, configuration
// validation, the tile layout, the staging plan of the state exchange) under AddressSanitizer + UBSan:
//   g++ -std=c++17 -x c++ -fsanitize=address,undefined -fno-sanitize-recover=all -I include -I /opt/rocm/include \
//       tests/host/host_logic_san.cpp -L/opt/rocm/lib -lamdhip64 -ldl -o host_logic_san && ./host_logic_san
// The kernels are not part of this build: the launchers are stubbed (no GPU is touched; cs_create fails
// with CS_ERR_DEVICE on a machine without one, which is one of the paths under test).
// Run by tests/test_host_sanitizers.py (-m "not gpu").
#define __HIP_PLATFORM_AMD__ 1
#include "../../gym_copter_amd/csrc/copterstep_api.hip"

#include <cassert>
#include <cinttypes>

namespace cs {
Tuning default_tuning() { return Tuning{98304u, 3670016u, 65536u}; }
bool launch_is_lean(const DevConst&, const DevState&) { return true; }
hipError_t launch_step(int, int, const DevConst&, const DevState&, const cs_step_io&, const Tuning&, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_step_many(int, int, const DevConst&, const DevState&, int, float*, float*, float*, uint8_t*, uint8_t*, int,
                            const PidConst*, double*, uint32_t, const Tuning&, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_export_state(int, const DevConst&, const DevState&, float*, uint8_t*, int32_t*, int32_t*, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_set_motors(int, const DevConst&, const DevState&, const float*, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_reset(int, int, const DevConst&, const DevState&, const uint8_t*, const float*, float*, double*, uint32_t,
                        const float*, int, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_set_perturbation(int, const DevState&, const uint8_t*, const float*, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_episode_stats(int, const DevConst&, const DevState&, double*, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_state_gather(int, const DevConst&, const DevState&, const StateArrays&, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_state_scatter(int, const DevConst&, const DevState&, const StateArrays&, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_serve(int, int, const DevConst&, const DevState&, const cs_serve_view&, hipStream_t) { return hipErrorUnknown; }
hipError_t serve_occupancy(int, int, const DevConst&, const DevState&, int*) { return hipErrorUnknown; }
hipError_t launch_serve_submit(const cs_serve_view&, uint32_t, const float*, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_serve_collect(const cs_serve_view&, int, float*, float*, uint8_t*, uint8_t*, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_serve_pid(const cs_serve_view&, uint32_t, uint32_t, const PidConst&, double*, uint32_t, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_serve_stop(uint32_t*, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_clock_probe(unsigned long long*, uint32_t, int, hipStream_t) { return hipErrorUnknown; }
}  // namespace cs

#define REQUIRE(cond)                                                        \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "REQUIRE failed line %d: %s\n", __LINE__, #cond); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

int main() {
  // ---- configuration defaults and validation (no device needed up to the device query) ----
  cs_config cfg;
  REQUIRE(cs_config_init(&cfg, CS_TASK_LANDER3D) == CS_OK);
  REQUIRE(cs_config_init(nullptr, 0) == CS_ERR_ARG && cs_config_init(&cfg, 99) == CS_ERR_ARG);
  REQUIRE(cs_config_init(&cfg, CS_TASK_HOVER3D) == CS_OK);
  cs_ctx* ctx = nullptr;
  cs_config bad = cfg;
  bad.struct_size += 4;
  REQUIRE(cs_create(&bad, &ctx) == CS_ERR_ABI && ctx == nullptr);
  bad = cfg;
  bad.num_envs = 0;
  REQUIRE(cs_create(&bad, &ctx) == CS_ERR_ARG);
  bad = cfg;
  bad.max_steps = 1 << 20;
  REQUIRE(cs_create(&bad, &ctx) == CS_ERR_ARG);
  bad = cfg;
  bad.action_arith = CS_ARITH_F32;
  bad.thrust_model = CS_THRUST_LIFT;
  REQUIRE(cs_create(&bad, &ctx) == CS_ERR_ARG && std::strstr(cs_last_error(), "float32 motor model") != nullptr);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    REQUIRE(cs_create(&cfg, &ctx) == CS_ERR_DEVICE && std::strstr(cs_last_error(), "no CPU fallback") != nullptr);
  REQUIRE(cs_step(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == CS_ERR_ARG);
  REQUIRE(cs_destroy(nullptr) == CS_OK && cs_comm_destroy(nullptr) == CS_OK);
  REQUIRE(cs_comm_create(nullptr, 1, 0, nullptr) == CS_ERR_ARG);

  // ---- constant folding: the live model and the Mars model, all tasks and state modes ----
  for (int task = 0; task < CS_TASK_COUNT; ++task) {
    for (int mode = CS_STATE_F32G; mode <= CS_STATE_F64; ++mode) {
      cs_ctx fake;
      std::memset(&fake, 0, sizeof fake);
      REQUIRE(cs_config_init(&fake.cfg, task) == CS_OK);
      fake.cfg.state_mode = mode;
      fake.cfg.seed = 0x0123456789ABCDEFull * (uint64_t)(task + 1);
      fake.cfg.env_id_base = ((int64_t)1 << 32) - 5;
      fake.layout = cs::make_layout(mode);
      const cs::DevConst c = make_const(&fake);
      REQUIRE(c.k_thrust < 0 && c.k_roll > 0 && c.k_yaw > 0 && c.dt == 0.01 && c.two_inv_M == 2.0 / 1.380);
      REQUIRE(std::isnan(c.reset_shaping) == !cs::task_is_lander(task));
      REQUIRE(c.gyro == 0 && c.act_f32 == 0 && c.key_force != c.key_action);
      // the two counters of the meta word at the default step limit (1000): 11 + 18 of its 29 counter bits
      REQUIRE(c.steps_bits == 11 && c.steps_mask == 2047 && c.ep_mask == (1u << 18) - 1 && c.ep_bits == 18);
      REQUIRE(c.f32_pi == 3.14159274101257324f && c.f32_LB == (float)(0.35 * 5e-3));
      fake.cfg.thrust_model = CS_THRUST_LIFT;
      fake.cfg.rotor_gyro = 1;
      fake.cfg.rho = 0.017;
      fake.cfg.G = 3.721;
      const cs::DevConst m = make_const(&fake);
      // Lift = 0.5*rho*S*C_L*(omega*L/2)^2 with S = 0.05*L*4: thrust per unit sum(m^2)
      const double ws = 15000 * kPi / 30, KL = 0.5 * 0.017 * (0.05 * 0.35 * 4) * 0.4 * (0.35 / 2) * (0.35 / 2) * ws * ws;
      REQUIRE(std::fabs(m.k_thrust + KL / 1.380) < 1e-12 * KL && std::fabs(m.k_roll - KL / 2) < 1e-12 * KL);
      REQUIRE(m.gyro == 1 && m.g_phi == 38e-4 / 2 * ws && m.G == 3.721);
      // ---- tile layout: the groups and rows of a tile follow each other without overlap ----
      const cs::Layout& L = fake.layout;
      const uint32_t gsz = 64u * 4u * L.word;
      REQUIRE(L.grp[0] == 0 && L.grp[1] == gsz && L.grp[2] == 2 * gsz && L.grp[3] == 3 * gsz && L.fe == 4 * gsz);
      REQUIRE(L.ret == L.fe + gsz && L.tile_bytes >= L.ret + 256u && L.tile_bytes % 256u == 0);
    }
  }

  // ---- the staging plan of cs_get_state / cs_set_state: requested arrays only, 256-byte aligned, disjoint ----
  {
    const bool want[Staging::kArrays] = {true, false, true, true, false, true, false, true, true};
    const Staging st(1000, want);
    REQUIRE(st.off[0] == 0 && st.size[0] == 96000 && st.off[2] == 96000 && st.off[3] == 96000 + 4096);
    REQUIRE(st.bytes % 256 == 0 && st.at<double>(1) == nullptr && st.arrays().status == nullptr);
    REQUIRE(st.arrays().ticks != nullptr && st.size[8] == 4000 && st.off[8] + 4096 == st.bytes);
    const bool none[Staging::kArrays] = {};
    REQUIRE(Staging(5, none).bytes == 0);
  }

  // ---- step-counter width by step limit: the saturated counter stays above max_steps + 1 ----
  REQUIRE(cs::steps_bits_for(1) == 3 && cs::steps_bits_for(2) == 3 && cs::steps_bits_for(3) == 4);
  REQUIRE(cs::steps_bits_for(1000) == 11 && cs::steps_bits_for(1022) == 11 && cs::steps_bits_for(1023) == 12);
  REQUIRE(cs::steps_bits_for((1 << 20) - 3) == 21 && cs::kMetaCounterBits - cs::kMetaStepsBitsMax == 8);

  // ---- seed mixing, environment overrides ----
  REQUIRE(splitmix64(0) == 0xE220A8397B1DCDAFull && splitmix64(1) != splitmix64(0x100000000ull));
  REQUIRE(env_u32("COPTERSTEP_SURELY_UNSET_VARIABLE") == 0);
  std::printf("host_logic_san: OK\n");
  return 0;
}
