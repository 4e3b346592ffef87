// model_6.hip -- instantiates the TrajOpt kernels of the internal model variant 6 (common.hpp: GUSTO_TO_*)
#include "launch.hpp"

int gusto_launch_init_m6(gusto_handle h, bool straight) { return launch_init<6>(h, straight); }
int gusto_launch_trajopt_m6(gusto_handle h, int mode, int max_iter) { return launch_trajopt<6>(h, mode, max_iter); }
