from hipla.ngstd import *    # noqa: F401,F403
from hipla.ngstd import Timer, TaskManager  # noqa: F401
