// model_5.hip -- instantiates the TrajOpt kernels of the internal model variant 5 (common.hpp: GUSTO_TO_*)
#include "launch.hpp"

int gusto_launch_init_m5(gusto_handle h, bool straight) { return launch_init<5>(h, straight); }
int gusto_launch_trajopt_m5(gusto_handle h, int mode, int max_iter) { return launch_trajopt<5>(h, mode, max_iter); }
