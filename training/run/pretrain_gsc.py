"""Same command line as the reference: ``python -m training.run.pretrain_gsc --model res8 --workspace W``."""
from howl_amd.training.run.pretrain_gsc import main

if __name__ == "__main__":
    main()
