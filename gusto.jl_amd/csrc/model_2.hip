// model_2.hip -- instantiates the SCP kernels for gusto_model_id 2
#include "launch.hpp"

int gusto_launch_init_m2(gusto_handle h, bool straight) { return launch_init<2>(h, straight); }
int gusto_launch_scp_m2(gusto_handle h, int mode, int max_iter, int force) { return launch_scp<2>(h, mode, max_iter, force); }
