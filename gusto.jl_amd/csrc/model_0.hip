// model_0.hip -- instantiates the SCP kernels for gusto_model_id 0
#include "launch.hpp"

int gusto_launch_init_m0(gusto_handle h, bool straight) { return launch_init<0>(h, straight); }
int gusto_launch_scp_m0(gusto_handle h, int mode, int max_iter, int force) { return launch_scp<0>(h, mode, max_iter, force); }
