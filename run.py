#!/usr/bin/env python
"""``python run.py --train|--test [--cfg ini]`` - same command line as the reference's run.py."""
from gan_sr_wind_field_amd.run import main

if __name__ == "__main__":
    main()
