from .constants import *  # noqa: F401,F403
from .utils import (wraptopi, flatten, list_to_variable_dict, variable_dict_to_list, normalize,  # noqa: F401
                    table_environment)
from .lie import SE2, SE3, SO2, SO3, trans_axis, rot_axis  # noqa: F401
