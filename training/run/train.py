"""Same command line as the reference: ``python -m training.run.train --model res8 --workspace W ...``."""
from howl_amd.training.run.train import main

if __name__ == "__main__":
    main()
